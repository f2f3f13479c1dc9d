"""Checkpoint location / loading for the wrappers.

Same directory convention as the reference: `MODEL_DATA_DIR` = $PIPELINE_3RDPARTY, else `<repo>/3rdparty`
(pose_pipeline/__init__.py:21-24), with the checkpoint paths the reference wrappers hard-code
(wrappers/mmpose.py:33-36, wrappers/videopose3d.py:52-54).  No checkpoint exists in the build or GPU
images, so POSEPIPE_SYNTHETIC_WEIGHTS=1 substitutes seeded synthetic parameters of the same
architecture (posepipeline_amd.models.synth) -- without it a missing file is an error, as in the reference.
"""
from __future__ import annotations

import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))


def model_data_dir() -> str:
    return os.environ.get("PIPELINE_3RDPARTY", os.path.join(_PKG, "..", "3rdparty"))


def load_state_dict(path: str, key_candidates=("state_dict", "model_pos")) -> dict:
    """torch checkpoint (.pth / .bin) -> {name: float32 numpy array}."""
    import torch
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt
    for k in key_candidates:
        if isinstance(ckpt, dict) and k in ckpt:
            sd = ckpt[k]
            break
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if hasattr(v, "detach")}


def get_state_dict(relpath: str, shapes: dict, seed: int, synth=None) -> dict:
    path = os.path.join(model_data_dir(), relpath)
    if os.path.exists(path):
        sd = load_state_dict(path)
        missing = [k for k in shapes if k not in sd]
        if missing:
            raise KeyError(f"{path}: missing parameters {missing[:5]}{'...' if len(missing) > 5 else ''}")
        return {k: np.asarray(sd[k], np.float32) for k in shapes}
    if os.environ.get("POSEPIPE_SYNTHETIC_WEIGHTS") == "1":
        if synth is not None:                      # model-specific generator (ViT: 1/fan_in linear layers)
            return synth(shapes, seed)
        from .models.synth import synth_state_dict
        return synth_state_dict(shapes, seed)
    raise FileNotFoundError(f"{path} (set POSEPIPE_SYNTHETIC_WEIGHTS=1 to run with seeded synthetic weights)")
