#!/usr/bin/env python
"""cProfile of Cascade.step on the bench workload: where does the host time between the GPU stages go?"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from posepipeline_amd import _lib  # noqa: E402
from posepipeline_amd.cascade import Cascade  # noqa: E402
from posepipeline_amd.models import faster_rcnn as fr, hrnet, synth  # noqa: E402
from posepipeline_amd.models import videopose3d as vp3d  # noqa: E402


def main():
    B = 32
    ctx = _lib.Context(0)
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    spec = hrnet.hrnet_w48_384x288()
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, 1080, 1920, chunk=B, max_persons=1)
    frames, gt = bench.synth_1080p(np.random.default_rng(3000), B, 1)
    d = ctx.malloc(frames.nbytes)
    ctx.h2d(d, frames)
    for _ in range(3):
        cas.step(None, frames_dev=(d, B), replay=gt)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(8):
        cas.step(None, frames_dev=(d, B), replay=gt)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
