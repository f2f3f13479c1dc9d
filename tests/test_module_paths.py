"""the drop-in is importable under the reference's own module paths (pose_pipeline.*), with the
reference's signatures, so `TrackingBbox.make` / `TopDownPerson.make` / `LiftingPerson.make` and the scripts need no
edit -- the repository root only has to precede a reference checkout on sys.path."""
import datetime
import inspect

import numpy as np


def test_reference_module_paths_resolve_to_the_drop_in():
    import pose_pipeline
    from pose_pipeline.wrappers.deep_sort_yolov4.parser import tracking_bounding_boxes
    from pose_pipeline.wrappers.mmpose import mmpose_top_down_person
    from pose_pipeline.wrappers.mmtrack import mmtrack_bounding_boxes
    from pose_pipeline.wrappers.videopose3d import process_videopose3d
    import pose_pipeline.pipeline as ref_pl
    import pose_pipeline.wrappers.mmpose as a
    import posepipeline_amd.pipeline as pl
    import posepipeline_amd.wrappers.mmpose as b
    assert a is b and ref_pl is pl                                       # one module object: one model cache, one schema
    assert pose_pipeline.TopDownPerson is pl.TopDownPerson and isinstance(pose_pipeline.MODEL_DATA_DIR, str)
    # call signatures of the boundary (wrappers/mmtrack.py:8, wrappers/mmpose.py:26, wrappers/videopose3d.py:19,
    # wrappers/deep_sort_yolov4/parser.py:18)
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E = inspect.Parameter.empty
    assert sig(mmtrack_bounding_boxes) == [("file_path", E), ("method", "tracktor")]
    assert sig(mmpose_top_down_person) == [("key", E), ("method", "HRNet_W48_COCO")]
    assert sig(process_videopose3d) == [("key", E), ("batch_size", 32), ("transform_coco", False)]
    assert [n for n, _ in sig(tracking_bounding_boxes)][:1] == ["file_path"]
    ns = {}
    exec("from pose_pipeline import *", ns)                              # scripts/process_h36m.py:1
    assert {"Video", "TrackingBbox", "PersonBbox", "TopDownPerson", "LiftingPerson"} <= set(ns)


def test_recipe_through_reference_paths(monkeypatch, tmp_path):
    """pose_pipeline.utils.standard_pipelines.lifting_pipeline drives the tables; the wrappers are looked up under the
    reference's module paths at make() time (stubs here: no GPU in this test)."""
    from pose_pipeline.utils.standard_pipelines import lifting_pipeline
    from pose_pipeline.utils.tracking import annotate_single_person
    import pose_pipeline.wrappers.mmpose as wmm
    import pose_pipeline.wrappers.mmtrack as wmt
    import pose_pipeline.wrappers.videopose3d as wvp
    from posepipeline_amd import djshim, pipeline as pl, video
    from posepipeline_amd.utils import standard_pipelines as sp
    assert annotate_single_person is sp.annotate_single_person
    djshim.reset()
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, np.zeros((5, 32, 48, 3), np.uint8), 30.0)
    vkey = {"video_project": "p", "filename": "q"}
    pl.Video.insert1({**vkey, "video": path, "start_time": datetime.datetime(2024, 5, 1)})
    rows = [[{"track_id": 7, "tlbr": np.array([1.0, 2, 11, 22]), "tlhw": np.array([1.0, 2, 10, 20]), "confidence": 0.9}]] * 5
    seen = []
    monkeypatch.setattr(wmt, "mmtrack_bounding_boxes", lambda file_path, method="tracktor": seen.append(method) or rows)
    monkeypatch.setattr(wmm, "mmpose_top_down_person", lambda key, method="x": np.ones((5, 17, 3), np.float32))
    monkeypatch.setattr(wvp, "process_videopose3d",
                        lambda key, **kw: {"keypoints_3d": np.zeros((5, 17, 3)), "keypoints_valid": [True] * 5})
    assert lifting_pipeline(vkey, tracking_method_name="MMTrack_deepsort", top_down_method_name="MMPose",
                            lifting_method_name="VideoPose3D") is True
    assert seen == ["deepsort"]                                          # tracking_method 5 = MMTrack_deepsort (pipeline.py:539-541)
    assert (pl.PersonBboxValid & vkey).fetch1("keep_tracks").tolist() == [7]
    assert (pl.LiftingPerson & vkey).fetch1("keypoints_3d").shape == (5, 17, 3)
    djshim.reset()
