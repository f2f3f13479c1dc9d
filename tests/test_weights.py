"""Checkpoint lookup / loading of the wrappers (posepipeline_amd/weights.py): same directory convention as the
reference (MODEL_DATA_DIR = $PIPELINE_3RDPARTY, pose_pipeline/__init__.py:21-24), torch checkpoints with the
`state_dict` / `model_pos` wrappers mmpose and VideoPose3D use, loud failure when a file is missing."""
import os

import numpy as np
import pytest
import torch

from posepipeline_amd import weights
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.models import videopose3d as vp3d


def test_checkpoint_roundtrip_and_errors(tmp_path, monkeypatch):
    monkeypatch.setenv("PIPELINE_3RDPARTY", str(tmp_path))
    monkeypatch.delenv("POSEPIPE_SYNTHETIC_WEIGHTS", raising=False)
    spec = hrnet.HRNetSpec(32, 17, 64, 64)
    shapes = hrnet.hrnet_param_shapes(spec)
    sd = synth.synth_state_dict(shapes, seed=3)
    rel = "mmpose/checkpoints/hrnet_test.pth"
    os.makedirs(tmp_path / "mmpose/checkpoints")
    # mmpose checkpoints: {"meta": ..., "state_dict": {...}} with extra buffers (num_batches_tracked) that must be ignored
    full = {k: torch.from_numpy(v) for k, v in sd.items()}
    full["backbone.bn1.num_batches_tracked"] = torch.tensor(7)
    torch.save({"meta": {"epoch": 210}, "state_dict": full}, tmp_path / rel)
    got = weights.get_state_dict(rel, shapes, seed=0)
    assert set(got) == set(shapes)
    for k in shapes:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], sd[k])
    # a checkpoint that lacks parameters of the architecture is an error, not a silent partial load
    partial = dict(list(full.items())[:10])
    torch.save({"state_dict": partial}, tmp_path / "mmpose/checkpoints/partial.pth")
    with pytest.raises(KeyError, match="missing parameters"):
        weights.get_state_dict("mmpose/checkpoints/partial.pth", shapes, seed=0)
    # VideoPose3D's .bin stores the weights under "model_pos"
    vshapes = vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec())
    vsd = synth.synth_state_dict(vshapes, seed=4)
    os.makedirs(tmp_path / "videopose3d")
    torch.save({"epoch": 80, "model_pos": {k: torch.from_numpy(v) for k, v in vsd.items()}}, tmp_path / "videopose3d/pretrained_h36m_detectron_coco.bin")
    got = weights.get_state_dict("videopose3d/pretrained_h36m_detectron_coco.bin", vshapes, seed=0)
    assert all(np.array_equal(got[k], vsd[k]) for k in vshapes)
    # no file, no opt-in: fail like the reference would
    with pytest.raises(FileNotFoundError, match="POSEPIPE_SYNTHETIC_WEIGHTS"):
        weights.get_state_dict("mmpose/checkpoints/absent.pth", shapes, seed=0)
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    syn = weights.get_state_dict("mmpose/checkpoints/absent.pth", shapes, seed=3)
    assert all(np.array_equal(syn[k], sd[k]) for k in shapes)
    assert os.path.samefile(weights.model_data_dir(), tmp_path)


def test_smooth_lifting_weights_are_metre_sized_contractions():
    """synth.smooth_lifting_state_dict (the lifting weights of the 3D tolerance tests): max-norm sensitivity <= 1 by construction
    (product of the layers' row sums) and measured, outputs metre-sized on screen-normalised inputs -- so "1e-3 mm" is a
    statement about the arithmetic, not about a weight scale."""
    from oracle import nets as onets
    from posepipeline_amd.models import videopose3d as vp3d
    sd = synth.smooth_lifting_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    bound = synth.max_norm_gain_bound(sd)
    assert 0.9 < bound <= 1.0
    model = onets.VideoPose3DRef(sd)
    rng = np.random.default_rng(0)
    t = np.arange(12)[:, None, None]
    kn = (rng.uniform(-0.5, 0.5, (1, 17, 2)) + 0.002 * t * rng.uniform(-1, 1, (1, 17, 2))).astype(np.float32)
    y = model.forward(onets.videopose3d_windows(kn, 121))
    assert 0.3 < np.abs(y).max() < 2.0 and y.std() > 0.1
    d = (rng.uniform(-1, 1, kn.shape) * 1e-3).astype(np.float32)
    y2 = model.forward(onets.videopose3d_windows(kn + d, 121))
    gain = np.abs(y2 - y).max() / np.abs(d).max()
    assert 0.05 < gain <= bound
    # the seeded He-normal weights the other tests use are NOT contractions (why they cannot carry a millimetre claim)
    assert synth.max_norm_gain_bound(synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)) > 10
