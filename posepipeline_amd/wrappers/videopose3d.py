"""Drop-in for pose_pipeline/wrappers/videopose3d.py:19-91 `process_videopose3d`.

Same signature, table reads (`TopDownPerson.keypoints`, `VideoInfo.height/width`) and return dict
({"keypoints_3d": (N,17,3) float64, "keypoints_valid": [True]*N}).  The reference builds one
edge-replicated 243-frame window per frame (ChunkedGenerator, :66-75) and runs
TemporalModelOptimized1f on the CPU in batches of `batch_size`; here the whole clip goes through the
dilated program on the GPU (pp_videopose3d_lift), which evaluates the same sums.  `batch_size` and
`transform_coco` are accepted and, like `transform_coco` in the reference, have no effect on the result.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .. import _lib, weights
from ..models import videopose3d as vp3d
from ..program import Net


@dataclass
class VideoPoseArgs:          # wrappers/videopose3d.py:10-16
    causal: bool = False
    architecture: str = "3,3,3,3,3"
    dropout: float = 0.25
    channels: int = 1024
    dense: bool = False


def normalize_screen_coordinates(X, w, h):
    """wrappers/videopose3d.py:26-33: map [0, w] to [-1, 1] preserving the aspect ratio."""
    assert X.shape[-1] == 2
    if w > h:
        return X / w * 2 - [1, h / w]
    else:
        return X / h * 2 - [w / h, 1]


_cache: dict = {}


def _model(device=0):
    if device not in _cache:
        args = VideoPoseArgs()
        fw = tuple(int(x) for x in args.architecture.split(","))
        spec = vp3d.VideoPose3DSpec(17, 2, 17, fw, args.channels)
        sd = weights.get_state_dict("videopose3d/pretrained_h36m_detectron_coco.bin", vp3d.videopose3d_param_shapes(spec),
                                    seed=3)
        ctx = _lib.Context(device)
        net = Net(ctx, vp3d.build_videopose3d_program(spec, sd), max_batch=4)
        _cache[device] = (ctx, net, spec)
    return _cache[device]


def lift(net, spec, keypoints_norm: np.ndarray) -> np.ndarray:
    """keypoints_norm (N, J, 2) -> (N, J_out, 3) float32 via pp_videopose3d_lift."""
    x = np.ascontiguousarray(keypoints_norm.astype("float32")).reshape(keypoints_norm.shape[0], -1)
    n = x.shape[0]
    out = np.zeros((n, spec.num_joints_out * 3), np.float32)
    _lib.check(net.ctx.lib.pp_videopose3d_lift(net.handle, net.prog.named["input"], net.prog.named["output"], _lib.ptr(x), n,
                                               x.shape[1], spec.num_joints_out * 3, spec.pad, _lib.ptr(out)),
               "pp_videopose3d_lift")
    return out.reshape(n, spec.num_joints_out, 3)


def process_videopose3d(key, batch_size=32, transform_coco=False):
    from ..pipeline import TopDownPerson, VideoInfo

    keypoints = (TopDownPerson & key).fetch1("keypoints")
    height, width = (VideoInfo & key).fetch1("height", "width")
    N = keypoints.shape[0]
    keypoints = normalize_screen_coordinates(keypoints[:, :, :2], width, height)
    valid_frames = np.arange(N)

    _, net, spec = _model()
    results = lift(net, spec, keypoints)

    keypoints_3d = np.zeros((N, 17, 3))
    keypoints_3d[valid_frames] = results
    keypoints_valid = [i in valid_frames.tolist() for i in np.arange(N)]
    return {"keypoints_3d": keypoints_3d, "keypoints_valid": keypoints_valid}
