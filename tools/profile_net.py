#!/usr/bin/env python
"""Per-op HIP-event profile of a backbone program (pp_net_profile): ms, TFLOP/s, share of the total.
usage: python tools/profile_net.py [w32|w48] [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from posepipeline_amd import _lib  # noqa: E402
from posepipeline_amd.models import hrnet, synth  # noqa: E402
from posepipeline_amd.program import Net  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "w32"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    if which in ("det", "roi"):
        from posepipeline_amd.models import faster_rcnn as fr
        sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
        prog = fr.build_image_program(sd, 640, 1088) if which == "det" else fr.build_roi_program(sd)
    elif which == "yolox":
        from posepipeline_amd.models import yolox
        prog = yolox.build_yolox_program(synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6), 800, 1344)
    elif which in ("yolo", "mars"):
        from posepipeline_amd.models import mars, yolov4
        if which == "yolo":
            prog = yolov4.build_yolov4_program(yolov4.synth_params(yolov4.yolov4_param_shapes(), seed=4))
        else:
            prog = mars.build_mars_program(yolov4.synth_params(mars.mars_param_shapes(), seed=5))
    else:
        spec = hrnet.hrnet_w32_256x192() if which == "w32" else hrnet.hrnet_w48_384x288()
        sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
        prog = hrnet.build_hrnet_program(spec, sd)
    ctx = _lib.Context(0)
    net = Net(ctx, prog, max_batch=batch)
    net.profile(batch)
    ms = np.median(np.stack([net.profile(batch) for _ in range(5)]), axis=0)
    tot = ms.sum()
    print(f"{which} batch {batch}: total {tot:.3f} ms, {prog.flops * batch / tot / 1e9:.2f} TFLOP/s over {len(ms)} ops")
    groups = {}
    for i, op in enumerate(prog.ops):
        bi, bo = prog.bufs[op.in_], prog.bufs[op.out]
        key = (bi[0], bi[1], op.cin, op.cout, op.kh, op.stride, op.up_log2)
        g = groups.setdefault(key, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += ms[i]
        g[2] += prog.op_flops[i] * batch
    print("  HxW  cin->cout k s up | count  ms  share  TFLOP/s")
    for key, (cnt, t, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        h, w, cin, cout, k, s, up = key
        print(f"  {h:3d}x{w:<3d} {cin:4d}->{cout:<4d} k{k} s{s} up{up} | {cnt:3d} {t:8.3f} {100 * t / tot:5.1f}% {fl / t / 1e9:7.2f}")


if __name__ == "__main__":
    main()
