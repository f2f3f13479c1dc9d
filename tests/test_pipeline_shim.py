"""Plumbing of the table layer on CPU (BASELINE.json configs[0] is "populate() plumbing, no GPU"):
keys flow Video -> TrackingBbox -> PersonBboxValid -> PersonBbox -> TopDownMethod -> TopDownPerson ->
LiftingMethod -> LiftingPerson with the reference's names; the GPU wrappers are replaced by recording
stubs here (the real ones are exercised by tests/test_gpu_pipeline.py)."""
import datetime

import numpy as np
import pytest

from posepipeline_amd import djshim, pipeline as pl


@pytest.fixture(autouse=True)
def clean():
    djshim.reset()
    yield
    djshim.reset()


def test_lookup_tables_match_reference_rows():
    assert (pl.TrackingBboxMethodLookup & {"tracking_method": 5}).fetch1("tracking_method_name") == "MMTrack_deepsort"
    assert (pl.TopDownMethodLookup & {"top_down_method": 0}).fetch1("top_down_method_name") == "MMPose"
    assert (pl.TopDownMethodLookup & {"top_down_method": 2}).fetch1("top_down_method_name") == "MMPoseHalpe"
    assert (pl.LiftingMethodLookup & {"lifting_method": 1}).fetch1("lifting_method_name") == "VideoPose3D"
    assert len(pl.TrackingBboxMethodLookup()) == 8 and len(pl.LiftingMethodLookup()) == 7
    # the reference's 13 rows keep their ids; ids >= 100 are this package's ViTPose extension (BASELINE.json configs[4])
    ref_rows = [r for r in pl.TopDownMethodLookup().fetch(as_dict=True) if r["top_down_method"] < 100]
    ext_rows = [r["top_down_method_name"] for r in pl.TopDownMethodLookup().fetch(as_dict=True) if r["top_down_method"] >= 100]
    assert len(ref_rows) == 13 and ext_rows == ["ViTPoseB", "ViTPoseL", "ViTPoseH"]


def test_primary_keys_follow_the_definitions():
    assert pl.Video.primary_key == ["video_project", "filename"]
    assert pl.TrackingBbox.primary_key == ["video_project", "filename", "tracking_method"]
    assert pl.PersonBbox.primary_key == ["video_project", "filename", "tracking_method", "video_subject_id"]
    assert pl.TopDownPerson.primary_key == pl.PersonBbox.primary_key + ["top_down_method"]
    assert pl.LiftingPerson.primary_key == pl.TopDownPerson.primary_key + ["lifting_method"]
    assert pl.TrackingBbox.heading[-2:] == ["tracks", "num_tracks"]
    assert pl.LiftingPerson.heading[-2:] == ["keypoints_3d", "keypoints_valid"]


def test_populate_chain_with_stub_wrappers(monkeypatch, tmp_path):
    from posepipeline_amd import video
    from posepipeline_amd.wrappers import mmpose as wmm, videopose3d as wvp
    calls = []
    frames = np.zeros((12, 48, 64, 3), np.uint8)
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, frames, 25.0)
    vkey = {"video_project": "p", "filename": "f"}
    pl.Video().insert1({**vkey, "video": path, "start_time": datetime.datetime(2024, 5, 1)})
    pl.VideoInfo().populate()
    assert (pl.VideoInfo & vkey).fetch1("fps") == 25.0 and len((pl.VideoInfo & vkey).fetch1("timestamps")) == 12

    def fake_track(file_path, method="tracktor"):
        calls.append(("track", file_path, method))
        out = []
        for t in range(12):
            fr = [{"track_id": 0, "tlbr": np.array([1.0, 2, 11, 22]), "tlhw": np.array([1.0, 2, 10, 20]), "confidence": 0.9}]
            if t % 4 == 0:
                fr.append({"track_id": 7, "tlbr": np.array([30.0, 2, 40, 22]), "tlhw": np.array([30.0, 2, 10, 20]), "confidence": 0.6})
            out.append(fr if t != 5 else [])
        return out

    import posepipeline_amd.wrappers as W
    import sys
    import types
    fake_mod = types.ModuleType("posepipeline_amd.wrappers.mmtrack")
    fake_mod.mmtrack_bounding_boxes = fake_track
    monkeypatch.setitem(sys.modules, "posepipeline_amd.wrappers.mmtrack", fake_mod)
    monkeypatch.setattr(W, "mmtrack", fake_mod, raising=False)
    tkey = {**vkey, "tracking_method": 5}
    pl.TrackingBboxMethod().insert1(tkey)
    pl.TrackingBbox().populate()
    assert calls[0] == ("track", path, "deepsort")
    assert (pl.TrackingBbox & tkey).fetch1("num_tracks") == 2
    # populate is idempotent: a key already in the target table is skipped
    pl.TrackingBbox().populate()
    assert len(calls) == 1
    with pytest.raises(djshim.DuplicateError):
        pl.TrackingBbox().insert1({**tkey, "tracks": [], "num_tracks": 0})
    # subject -1 is excluded by PersonBbox.key_source (pipeline.py:705-707)
    pl.PersonBboxValid().insert1({**tkey, "video_subject_id": -1, "keep_tracks": [7]})
    pl.PersonBboxValid().insert1({**tkey, "video_subject_id": 0, "keep_tracks": [0]})
    pl.PersonBbox().populate()
    assert len(pl.PersonBbox()) == 1
    bbox, present = (pl.PersonBbox & tkey).fetch1("bbox", "present")
    assert present.all() and np.array_equal(bbox[5], [1, 2, 10, 20])      # single dropout back-filled
    pl.DetectedFrames().populate()
    df = (pl.DetectedFrames & tkey).fetch1()
    assert df["frames_detected"] == 11 and df["frames_missed"] == 1 and abs(df["mean_other_people"] - 3 / 12) < 1e-12

    def fake_2d(key, method="HRNet_W48_COCO"):
        calls.append(("2d", dict(key), method))
        n = len((pl.PersonBbox & key).fetch1("bbox"))
        return np.ones((n, 17, 3), np.float32)

    def fake_3d(key, batch_size=32, transform_coco=False):
        calls.append(("3d", dict(key)))
        kp = (pl.TopDownPerson & key).fetch1("keypoints")
        return {"keypoints_3d": np.zeros((kp.shape[0], 17, 3)), "keypoints_valid": [True] * kp.shape[0]}

    monkeypatch.setattr(wmm, "mmpose_top_down_person", fake_2d)
    monkeypatch.setattr(wvp, "process_videopose3d", fake_3d)
    pkey = {**tkey, "video_subject_id": 0, "top_down_method": 0}
    pl.TopDownMethod().insert1(pkey)
    pl.TopDownPerson().populate()
    assert calls[-1][0] == "2d" and calls[-1][2] == "HRNet_W48_COCO" and calls[-1][1]["top_down_method"] == 0
    pl.TopDownMethod().insert1({**pkey, "top_down_method": 4})             # OpenPose row: not on this path
    with pytest.raises(Exception, match="not implemented"):
        pl.TopDownPerson().populate()
    errs = pl.TopDownPerson().populate(suppress_errors=True)
    assert len(errs) == 1
    lkey = {**pkey, "lifting_method": 1}
    pl.LiftingMethod().insert1(lkey)
    pl.LiftingPerson().populate(lkey)
    assert calls[-1][0] == "3d" and (pl.LiftingPerson & lkey).fetch1("keypoints_3d").shape == (12, 17, 3)
    assert pl.LiftingPerson.joint_names()[0] == "Hip (root)" and len(pl.TopDownPerson.joint_names()) == 17


def test_lifting_pipeline_recipe_with_stub_wrappers(monkeypatch, tmp_path):
    """utils/standard_pipelines.py:110-164 on the shim: one call takes a video to LiftingPerson."""
    import sys
    import types
    from posepipeline_amd import video
    from posepipeline_amd.utils import standard_pipelines as sp
    from posepipeline_amd.wrappers import mmpose as wmm, videopose3d as wvp
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, np.zeros((9, 32, 48, 3), np.uint8), 30.0)
    vkey = {"video_project": "p", "filename": "g"}
    pl.Video.insert1({**vkey, "video": path, "start_time": datetime.datetime(2024, 5, 1)})
    one = [[{"track_id": 3, "tlbr": np.array([1.0, 2, 11, 22]), "tlhw": np.array([1.0, 2, 10, 20]), "confidence": 0.9}]] * 9
    fake = types.ModuleType("posepipeline_amd.wrappers.mmtrack")
    fake.mmtrack_bounding_boxes = lambda file_path, method="tracktor": one
    monkeypatch.setitem(sys.modules, "posepipeline_amd.wrappers.mmtrack", fake)
    import posepipeline_amd.wrappers as W
    monkeypatch.setattr(W, "mmtrack", fake, raising=False)
    monkeypatch.setattr(wmm, "mmpose_top_down_person", lambda key, method="x": np.ones((9, 17, 3), np.float32))
    monkeypatch.setattr(wvp, "process_videopose3d",
                        lambda key, **kw: {"keypoints_3d": np.zeros((9, 17, 3)), "keypoints_valid": [True] * 9})
    names = dict(tracking_method_name="MMTrack_deepsort", top_down_method_name="MMPose", lifting_method_name="VideoPose3D")
    # the DEFAULTS are the reference's (utils/standard_pipelines.py:112-114) and behave like there: "MMpose" is not a row of
    # TopDownMethodLookup, so a call that relies on it fails in the lookup; "GastNet" is a row whose wrapper is out of scope
    import inspect
    sig = inspect.signature(sp.lifting_pipeline).parameters
    assert (sig["tracking_method_name"].default, sig["top_down_method_name"].default, sig["lifting_method_name"].default) == \
        ("DeepSortYOLOv4", "MMpose", "GastNet")
    assert inspect.signature(sp.tracking_pipeline).parameters["tracking_method_name"].default == "DeepSortYOLOv4"
    with pytest.raises(Exception, match="fetch1"):
        sp.top_down_pipeline(vkey, tracking_method_name="MMTrack_deepsort")          # default "MMpose": unknown method name
    with pytest.raises(Exception, match="fetch1"):
        sp.tracking_pipeline(vkey, tracking_method_name="NoSuchTracker")
    with pytest.raises(Exception, match="not implemented"):
        sp.lifting_pipeline(vkey, tracking_method_name="MMTrack_deepsort", top_down_method_name="MMPose")   # default "GastNet"
    (pl.LiftingMethod & {**vkey, "lifting_method": 0}).delete()          # the row that call registered (as in the reference)
    assert sp.lifting_pipeline(vkey, **names) is True
    assert (pl.PersonBboxValid & vkey).fetch1("keep_tracks").tolist() == [3]       # auto-annotated: one identity
    assert len(pl.BestDetectedFrames & vkey) == 1 and (pl.BestDetectedFrames & vkey).fetch1("KEY")["video_subject_id"] == 0
    assert len(pl.TopDownPerson & vkey) == 1 and len(pl.LiftingPerson & vkey) == 1
    assert (pl.LiftingMethod & vkey).fetch1("lifting_method") == 1
    # a second call is a no-op that still reports success
    assert sp.lifting_pipeline(vkey, **names) is True
    # two identities: no automatic annotation -> the recipe waits
    vkey2 = {"video_project": "p", "filename": "h"}
    pl.Video.insert1({**vkey2, "video": path, "start_time": datetime.datetime(2024, 5, 1)})
    two = [one[0] + [{"track_id": 4, "tlbr": np.array([30.0, 2, 40, 22]), "tlhw": np.array([30.0, 2, 10, 20]), "confidence": 0.8}]] * 9
    fake.mmtrack_bounding_boxes = lambda file_path, method="tracktor": two
    assert sp.lifting_pipeline(vkey2, **names) is False or sp.lifting_pipeline(vkey2, **names) == []
    assert len(pl.PersonBbox & vkey2) == 0
