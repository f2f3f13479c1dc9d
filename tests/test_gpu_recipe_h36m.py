"""The reference's own driver call (scripts/process_h36m.py:15):
    top_down_pipeline(k, top_down_method_name="MMPoseHalpe", tracking_method_name='DeepSortYOLOv4')
on the table shim with the GPU wrappers: DeepSortYOLOv4 tracking -> (annotation of the subject, as a user would do in
the reference's GUI) -> PersonBbox -> Halpe-136 top-down keypoints."""
import datetime

import numpy as np
import pytest

from tests.test_gpu_detector import synth_frame

pytestmark = pytest.mark.gpu


def test_process_h36m_recipe(ctx, tmp_path, monkeypatch):
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import djshim, pipeline as pl, video
    from posepipeline_amd.utils.standard_pipelines import top_down_pipeline
    from posepipeline_amd.wrappers import mmpose as wmm
    from posepipeline_amd.wrappers.deep_sort_yolov4 import parser
    djshim.reset()
    rng = np.random.default_rng(21)
    frames = np.stack([synth_frame(rng, 192, 256) for _ in range(6)])
    path = str(tmp_path / "S1_walking.ppvid")
    video.write_ppvid(path, frames, fps=50.0)
    key = {"video_project": "h36m", "filename": "S1_walking"}
    pl.Video().insert1({**key, "video": path, "start_time": datetime.datetime(2024, 1, 1)})
    # a smaller backbone keeps the test quick; the method row, wrapper and 136-joint head are the real ones
    from posepipeline_amd.models import hrnet
    spec_fn = lambda k: hrnet.HRNetSpec(32, k, 128, 96)
    monkeypatch.setitem(wmm._METHODS, "HRNet_W48_HALPE", (spec_fn,) + wmm._METHODS["HRNet_W48_HALPE"][1:])
    wmm._cache.clear()

    res = top_down_pipeline(key, top_down_method_name="MMPoseHalpe", tracking_method_name="DeepSortYOLOv4")
    tkey = {**key, "tracking_method": 0}
    tracks, num_tracks = (pl.TrackingBbox & tkey).fetch1("tracks", "num_tracks")
    assert len(tracks) == 6 and num_tracks >= 1
    assert all(set(t) == {"track_id", "tlhw", "tlbr", "time_since_update"} for fr in tracks for t in fr)
    if num_tracks != 1:
        # several identities: the recipe waits for the annotation, exactly like the reference (standard_pipelines.py:78-86)
        assert not res and len(pl.PersonBbox & tkey) == 0
        ids = [t["track_id"] for t in tracks[-1]]
        pl.PersonBboxValid().insert1({**tkey, "video_subject_id": 0, "keep_tracks": [ids[0]]})
        res = top_down_pipeline(key, top_down_method_name="MMPoseHalpe", tracking_method_name="DeepSortYOLOv4")
    assert res and len(res) == 1
    pkey = res[0]
    assert pkey["top_down_method"] == 2 and pkey["tracking_method"] == 0
    kp = (pl.TopDownPerson & pkey).fetch1("keypoints")
    bbox, present = (pl.PersonBbox & pkey).fetch1("bbox", "present")
    assert kp.shape == (6, 136, 3) and bbox.shape == (6, 4)
    assert all(kp[i].any() == bool(present[i]) for i in range(6))
    wmm._cache.clear()
    parser._cache.clear()
