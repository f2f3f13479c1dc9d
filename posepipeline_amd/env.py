"""Process environment of the pipeline: the package-level helpers the reference's entry scripts call first.

Mirrors pose_pipeline/env.py of the reference -- same names, arguments and error behaviour -- for a package whose hot
path has no third-party checkout to import:

  * `add_path` (env.py:9-27): context manager that puts one path or a list of paths at the front of `sys.path` and takes
    them out again on exit (a path that is already gone is not an error).
  * `set_environmental_variables(pose_project_dir=None)` (env.py:30-65): resolves the project directory (argument, else
    `dj.config["custom"]["pose_project_dir"]` through `paths.get_pose_project_dir`), asserts it is a directory and exports
    the `*_PATH` registry below -- the reference's variable names and the reference's sub-directories.  The reference
    asserts that EVERY checkout exists, because its wrappers import code from them (`wrappers/videopose3d.py:38` does
    `add_path(os.environ["VIDEOPOSE3D_PATH"])`).  Here the cascade is native (VideoPose3D is `csrc/lifting.hip`, not an
    import), so no key is needed by the path: a checkout that exists is exported, one that does not is skipped and listed in
    the return value; `strict=True` restores the reference's assertion for callers that mix in out-of-scope wrappers.
  * `pytorch_memory_limit(frac=0.5)` / `tensorflow_memory_limit()` (env.py:95-118): the reference caps the frameworks'
    caching allocators so that both fit one GPU.  This package's device memory is its own arenas (`pp_ctx_create`,
    hipMalloc), which neither call governs; they are applied to the frameworks when those are importable and see a GPU, and are
    harmless no-ops otherwise (ROCm image without TensorFlow; CPU-only container).
  * `jax_memory_limit()` (env.py:89-92).

`download_git_dependencies` (env.py:68-86, clones the third-party repositories) has no counterpart: nothing is cloned.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

from .paths import get_pose_project_dir

# variable -> sub-directory of the project directory (pose_pipeline/env.py:42-63).  Only VIDEOPOSE3D_PATH names a stage
# of the hot path, and that stage is native here; the others belong to wrappers outside SURVEY.md section 8.
ENV_PATHS = {
    "OPENPOSE_PATH": "openpose",
    "OPENPOSE_PYTHON_PATH": "openpose/python",
    "EXPOSE_PATH": "expose",
    "CENTERHMR_PATH": "CenterHMR",
    "GAST_PATH": "GAST-Net-3DPoseEstimation",
    "POSEFORMER_PATH": "PoseFormer",
    "VIBE_PATH": "VIBE",
    "MEVA_PATH": "MEVA",
    "PARE_PATH": "PARE",
    "PIXIE_PATH": "PIXIE",
    "HUMOR_PATH": "humor/humor",
    "FAIRMOT_PATH": "FairMOT/src/lib",
    "DCNv2_PATH": "DCNv2/DCN",
    "TRANSTRACK_PATH": "TransTrack",
    "PROHMR_PATH": "ProHMR",
    "TRADES_PATH": "TraDeS/src/lib",
    "RIE_PATH": "Pose3D-RIE",
    "VIDEOPOSE3D_PATH": "VideoPose3D",
    "POSEAUG_PATH": "PoseAug",
    "HYBRIDIK_PATH": "HybrIK",
}


class add_path:
    """`with add_path(p):` -- p (a path or a list of paths) leads sys.path inside the block."""

    def __init__(self, path):
        self.path = list(path) if isinstance(path, (list, tuple)) else [path]

    def __enter__(self):
        for p in self.path:
            sys.path.insert(0, p)
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        for p in self.path:
            if p in sys.path:
                sys.path.remove(p)
        return False


def set_environmental_variables(pose_project_dir=None, strict=False):
    """Export the `*_PATH` variables of the checkouts found under the project directory.

    pose_project_dir: root directory of the non-package dependencies, ending in "/" like the reference's (a missing
    trailing separator is added instead of silently producing `<dir>openpose`); None -> `get_pose_project_dir()`.
    Returns the list of variables that were NOT set because their directory is absent (always empty when strict=True,
    which asserts instead, like the reference)."""
    if not pose_project_dir:
        pose_project_dir = get_pose_project_dir()
    assert Path(pose_project_dir).is_dir(), f"Could not find pose project directory: {pose_project_dir}"
    root = str(pose_project_dir)
    if not root.endswith(("/", os.sep)):
        root += "/"
    skipped = []
    for var, sub in ENV_PATHS.items():
        path = root + sub
        if Path(path).exists():
            os.environ[var] = path
        elif strict:
            raise AssertionError(f"Could not find path {path}")
        else:
            skipped.append(var)
    import platform
    if "Ubuntu" in platform.version():          # env.py:69-71: off-screen rendering back end of the (out-of-scope) renderers
        os.environ["PYOPENGL_PLATFORM"] = "egl"
    return skipped


def jax_memory_limit():
    os.environ["XLA_PYTHON_CLIENT_PREALLOCATE"] = "false"


def pytorch_memory_limit(frac=0.5):
    """Cap PyTorch's caching allocator at `frac` of device 0 (env.py:95-100).  No effect on this package's own arenas;
    returns True when the cap was applied, False when there is no torch / no GPU to apply it to."""
    try:
        import torch
    except ImportError:
        return False
    if not torch.cuda.is_available():
        return False
    torch.cuda.set_per_process_memory_fraction(frac, 0)
    torch.cuda.empty_cache()
    return True


def tensorflow_memory_limit():
    """Switch TensorFlow to on-demand GPU memory growth (env.py:103-118).  TensorFlow is not a dependency of this package
    (tracking_method 0's networks are native programs): without it this is a no-op that returns False."""
    try:
        import tensorflow as tf
    except ImportError:
        return False
    gpus = tf.config.list_physical_devices("GPU")
    try:
        for gpu in gpus:
            tf.config.experimental.set_memory_growth(gpu, True)
    except RuntimeError as e:                   # growth must be chosen before the GPUs are initialised
        print(e)
    return bool(gpus)
