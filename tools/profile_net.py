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
    kinds = net.conv_kinds()
    elems = lambda b: prog.bufs[b][0] * prog.bufs[b][1] * prog.bufs[b][2]
    groups = {}
    for i, op in enumerate(prog.ops):
        bi, bo = prog.bufs[op.in_], prog.bufs[op.out]
        key = (bi[0], bi[1], op.cin, op.cout, op.kh, op.stride, op.up_log2, int(kinds[i]), op.type)
        g = groups.setdefault(key, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += ms[i]
        g[2] += prog.op_flops[i] * batch
        # algorithmic bytes: every operand read once, the output written once (float32)
        ins = sum(elems(b) for b in (op.in_, op.res1, op.res2, op.in2, op.in3) if b >= 0)
        out = bo[0] * bo[1] * (op.cout if op.type == 1 else bo[2])
        g[3] += 4.0 * (batch * (ins + out) + (op.kh * op.kw * op.cin * op.cout if op.type == 1 else 0))
    print("  HxW  cin->cout k s up family | count  ms  share  TFLOP/s  algorithmic TB/s")
    fam = {0: "other", 1: "fp32 ", 2: "split"}
    for key, (cnt, t, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        h, w, cin, cout, k, s, up, kind, typ = key
        print(f"  {h:3d}x{w:<3d} {cin:4d}->{cout:<4d} k{k} s{s} up{up} {fam[kind]} | {cnt:3d} {t:8.3f} {100 * t / tot:5.1f}% {fl / t / 1e9:7.2f} {by / t / 1e9:7.2f}")
    for kind in (2, 1, 0):
        sel = [(t, fl, by) for key, (cnt, t, fl, by) in groups.items() if key[7] == kind]
        if sel:
            t, fl, by = (sum(x) for x in zip(*sel))
            print(f"  {fam[kind]}: {t:8.3f} ms {100 * t / tot:5.1f}%  {fl / t / 1e9:7.2f} TFLOP/s  {by / t / 1e9:6.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
