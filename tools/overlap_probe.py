#!/usr/bin/env python
"""Probe: do the detector and the pose backbone fill each other's tail waves when they run concurrently from two
contexts (two stream sets) on one GPU?  Prints sequential vs concurrent wall time for K chunk-steps of each."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import _lib, ops  # noqa: E402
from posepipeline_amd.models import faster_rcnn as fr, hrnet, synth  # noqa: E402
from posepipeline_amd.program import Net  # noqa: E402


def main():
    B, K = 32, 6
    ctx_a, ctx_b = _lib.Context(0), _lib.Context(0)
    det = fr.Detector(ctx_a, synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2), 1080, 1920, max_frames=B)
    spec = hrnet.hrnet_w48_384x288()
    net = Net(ctx_b, hrnet.build_hrnet_program(spec, synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)), max_batch=2 * B)
    td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17))
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 255, (B, 1080, 1920, 3), dtype=np.uint8)
    da, db = ctx_a.malloc(frames.nbytes), ctx_b.malloc(frames.nbytes)
    ctx_a.h2d(da, frames)
    ctx_b.h2d(db, frames)
    fidx = np.arange(B, dtype=np.int32)
    boxes = np.tile(np.array([[800.0, 300.0, 300.0, 600.0]]), (B, 1))

    def run_det():
        for _ in range(K):
            det.run(None, frames_dev=(da, B))

    def run_pose():
        for _ in range(K):
            td.run(db, fidx, boxes, frames_dev_shape=(B, 1080, 1920))

    run_det(); run_pose()
    t0 = time.perf_counter(); run_det(); t_det = time.perf_counter() - t0
    t0 = time.perf_counter(); run_pose(); t_pose = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = threading.Thread(target=run_det)
    th.start()
    run_pose()
    th.join()
    t_both = time.perf_counter() - t0
    print(f"detector {t_det / K * 1e3:.1f} ms/step, pose {t_pose / K * 1e3:.1f} ms/step, sequential {(t_det + t_pose) / K * 1e3:.1f}, "
          f"concurrent {t_both / K * 1e3:.1f} ms/step ({(t_det + t_pose) / t_both:.3f}x)")


if __name__ == "__main__":
    main()
