"""the reference's own entry script runs against the drop-in without an edit.

`scripts/process_h36m.py:1-15` of the reference is, statement by statement: import the package, export the checkout paths
(`set_environmental_variables()`), star-import the tables, import the recipe, cap the two frameworks' GPU memory through the
`pose_pipeline.env` ATTRIBUTE, select the videos of one project, run `top_down_pipeline` per key with the Halpe method and the
DeepSortYOLOv4 tracker.  The sequence is restated below (never copied) and executed in a fresh interpreter through
`import pose_pipeline`; the two GPU wrappers are replaced by recording stubs (CPU test -- the real ones run in
tests/test_gpu_recipe_h36m.py).  A second test runs the table module on the `POSEPIPE_USE_DATAJOINT=1` branch
(posepipeline_amd/pipeline.py:18-19) against a stand-in `datajoint` module.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env=None, cwd=None):
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=e, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


SETUP = """
    import datetime, json, os, sys, types
    import numpy as np
    project = sys.argv[1] if len(sys.argv) > 1 else os.environ["TEST_PROJECT_DIR"]
    from posepipeline_amd import djshim, video
    djshim.config["custom"]["pose_project_dir"] = project          # what a user's dj_local_conf.json carries
    calls = []
    def fake_track(file_path, **kw):
        calls.append(["track", os.path.basename(file_path)])
        return [[{"track_id": 3, "tlbr": np.array([1.0, 2, 11, 22]), "tlhw": np.array([1.0, 2, 10, 20]), "confidence": 0.8}]] * 6
    def fake_topdown(key, method="HRNet_W48_COCO"):
        calls.append(["topdown", method])
        return np.ones((6, 136, 3), np.float32)
    import posepipeline_amd.wrappers.deep_sort_yolov4.parser as parser
    import posepipeline_amd.wrappers.mmpose as wmm
    parser.tracking_bounding_boxes = fake_track
    wmm.mmpose_top_down_person = fake_topdown
    import posepipeline_amd.pipeline as pl
    for name in ("a.ppvid", "b.ppvid"):
        path = os.path.join(project, name)
        video.write_ppvid(path, np.zeros((6, 32, 48, 3), np.uint8), 30.0)
        pl.Video.insert1({"video_project": "h36m", "filename": name, "video": path, "start_time": datetime.datetime(2024, 5, 1)})
    pl.Video.insert1({"video_project": "other", "filename": "c", "video": path, "start_time": datetime.datetime(2024, 5, 1)})
"""

# the reference script's statements, restated (scripts/process_h36m.py:1-15)
SCRIPT = """
    import pose_pipeline
    skipped = pose_pipeline.set_environmental_variables()
    from pose_pipeline import *
    from pose_pipeline.utils.standard_pipelines import top_down_pipeline

    pose_pipeline.env.pytorch_memory_limit()
    pose_pipeline.env.tensorflow_memory_limit()

    VIDEO_PROJECT = "h36m"
    keys = (Video & f'video_project="{VIDEO_PROJECT}"').fetch('KEY')

    for k in keys:
        top_down_pipeline(k, top_down_method_name="MMPoseHalpe", tracking_method_name='DeepSortYOLOv4')
"""

REPORT = """
    print(json.dumps({"calls": calls, "n_keys": len(keys), "topdown_rows": len(TopDownPerson()),
                      "best": len(BestDetectedFrames()), "vp3d": os.environ.get("VIDEOPOSE3D_PATH"),
                      "openpose": os.environ.get("OPENPOSE_PATH"), "skipped": skipped,
                      "shape": list((TopDownPerson & keys[0]).fetch1("keypoints").shape)}))
"""


def test_reference_entry_script_sequence_runs_unchanged(tmp_path):
    os.makedirs(tmp_path / "VideoPose3D")                                 # the one checkout this "installation" has
    out = _run(SETUP + SCRIPT + REPORT, env={"TEST_PROJECT_DIR": str(tmp_path) + "/"})
    rep = json.loads(out.strip().splitlines()[-1])
    assert rep["n_keys"] == 2 and rep["topdown_rows"] == 2 and rep["best"] == 2
    assert rep["calls"] == [["track", "a.ppvid"], ["topdown", "HRNet_W48_HALPE"], ["track", "b.ppvid"], ["topdown", "HRNet_W48_HALPE"]]
    assert rep["shape"] == [6, 136, 3]
    assert rep["vp3d"] == str(tmp_path) + "/VideoPose3D" and rep["openpose"] is None
    assert "OPENPOSE_PATH" in rep["skipped"] and "VIDEOPOSE3D_PATH" not in rep["skipped"]


def test_env_api_semantics(tmp_path, monkeypatch):
    import pose_pipeline
    from pose_pipeline import add_path, set_environmental_variables
    from pose_pipeline.paths import get_pose_project_dir
    from pose_pipeline.utils.paths import find_full_path
    from posepipeline_amd import djshim, env
    assert pose_pipeline.env is env and pose_pipeline.BestDetectedFrames.__name__ == "BestDetectedFrames"
    # add_path: one path or a list, first on sys.path inside, gone outside, tolerant of a path removed meanwhile (env.py:9-27)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    with add_path(a):
        assert sys.path[0] == a
    assert a not in sys.path
    with add_path([a, b]):
        assert sys.path[:2] == [b, a]
        sys.path.remove(a)
    assert a not in sys.path and b not in sys.path
    # project directory: must be a directory (argument or dj.config custom entry), same assertion text as the reference
    with pytest.raises(AssertionError, match="Could not find pose project directory"):
        set_environmental_variables(str(tmp_path / "nope") + "/")
    monkeypatch.setitem(djshim.config["custom"], "pose_project_dir", str(tmp_path) + "/")
    assert get_pose_project_dir() == str(tmp_path) + "/"
    for k in env.ENV_PATHS:
        monkeypatch.delenv(k, raising=False)
    os.makedirs(tmp_path / "humor" / "humor")
    skipped = set_environmental_variables()
    assert os.environ["HUMOR_PATH"] == str(tmp_path) + "/humor/humor" and "HUMOR_PATH" not in skipped
    assert len(skipped) == len(env.ENV_PATHS) - 1 and "VIDEOPOSE3D_PATH" not in os.environ
    with pytest.raises(AssertionError, match="Could not find path .*openpose"):
        set_environmental_variables(strict=True)                          # the reference's behaviour: every checkout must exist
    assert set_environmental_variables(str(tmp_path)) == skipped          # missing trailing separator is tolerated
    # the registry is the reference's (env.py:42-63): 20 variables, these spellings
    assert len(env.ENV_PATHS) == 20 and env.ENV_PATHS["FAIRMOT_PATH"] == "FairMOT/src/lib" and "DCNv2_PATH" in env.ENV_PATHS
    # memory caps: no GPU / no TensorFlow here -> no-ops that say so
    assert env.pytorch_memory_limit() is False and env.tensorflow_memory_limit() is False
    env.jax_memory_limit()
    assert os.environ["XLA_PYTHON_CLIENT_PREALLOCATE"] == "false"
    # find_full_path (utils/paths.py:9-33)
    (tmp_path / "x.bin").write_bytes(b"1")
    assert find_full_path([str(tmp_path / "nope"), str(tmp_path)], "x.bin") == tmp_path / "x.bin"
    assert find_full_path(str(tmp_path), str(tmp_path / "x.bin")) == tmp_path / "x.bin"
    with pytest.raises(FileNotFoundError):
        find_full_path(str(tmp_path), "y.bin")


FAKE_DJ = """
    import sys, types
    from posepipeline_amd import djshim
    dj = types.ModuleType("datajoint")
    log = []
    for n in ("Manual", "Lookup", "Computed", "config"):
        setattr(dj, n, getattr(djshim, n))
    def schema(name, *a, **kw):
        log.append(name)
        return djshim.schema(name, *a, **kw)
    dj.schema = schema
    dj.config["custom"]["database.prefix"] = "lab_"
    sys.modules["datajoint"] = dj
"""


def test_real_datajoint_branch_is_taken(tmp_path):
    """POSEPIPE_USE_DATAJOINT=1: the tables are declared on whatever `import datajoint` yields, under the schema name
    `<database.prefix>pose_pipeline` (pipeline.py:15-20 of the reference), and the recipes run on it."""
    out = _run(FAKE_DJ + """
    import posepipeline_amd.pipeline as pl
    import posepipeline_amd.paths as paths
    assert pl.dj is dj and paths._dj_config() is dj.config
    assert log == ["lab_pose_pipeline"], log
    assert issubclass(pl.TrackingBbox, dj.Computed) and issubclass(pl.Video, dj.Manual)
    assert len(pl.TopDownMethodLookup()) == 16
    print("ok")
    """, env={"POSEPIPE_USE_DATAJOINT": "1"})
    assert out.strip().endswith("ok")
    # without the switch the stand-in is never imported
    out = _run(FAKE_DJ + """
    import posepipeline_amd.pipeline as pl
    assert pl.dj is djshim and log == []
    print("ok")
    """, env={"POSEPIPE_USE_DATAJOINT": "0"})
    assert out.strip().endswith("ok")
