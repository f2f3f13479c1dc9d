#!/usr/bin/env python
"""A whole backbone on the bf16-split conv kernels vs the same net on the exact fp32-MFMA kernels: output difference and time.
usage: python tools/split_net_check.py [w32|w48|det|roi] [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from posepipeline_amd import _lib as L  # noqa: E402
from posepipeline_amd.models import hrnet, synth  # noqa: E402
from posepipeline_amd.program import Net  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "w48"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out_name = "output"
    if which in ("det", "roi"):
        from posepipeline_amd.models import faster_rcnn as fr
        sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
        prog = fr.build_image_program(sd, 640, 1088) if which == "det" else fr.build_roi_program(sd)
    else:
        spec = hrnet.hrnet_w32_256x192() if which == "w32" else hrnet.hrnet_w48_384x288()
        sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
        prog = hrnet.build_hrnet_program(spec, sd)
    ctx = L.Context(0)
    net = Net(ctx, prog, max_batch=batch)
    rng = np.random.default_rng(0)
    in_name = "input" if "input" in prog.named else "roi_in"
    pin = prog.named[in_name]
    x = rng.standard_normal((batch,) + tuple(prog.bufs[pin])).astype(np.float32)
    names = [k for k in prog.named if k != in_name]
    res = {}
    for v in (3, -1):
        L.check(ctx.lib.pp_conv_variant(v), "variant")
        res[v] = {nm: net.forward(x, in_name=in_name, out_name=nm) for nm in names}
        net.profile(batch)
        ms = np.median(np.stack([net.profile(batch) for _ in range(5)]), axis=0)
        print(f"variant {v}: {ms.sum():.3f} ms serial, {prog.flops * batch / ms.sum() / 1e9:.1f} TFLOP/s")
        if v == -1 or True:
            groups = {}
            for i, op in enumerate(prog.ops):
                bi = prog.bufs[op.in_]
                key = (bi[0], bi[1], op.cin, op.cout, op.kh, op.stride, op.up_log2)
                g = groups.setdefault(key, [0, 0.0, 0.0])
                g[0] += 1
                g[1] += ms[i]
                g[2] += prog.op_flops[i] * batch
            for key, (cnt, t, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:14]:
                h, w, cin, cout, k, s, up = key
                print(f"    {h:3d}x{w:<4d} {cin:4d}->{cout:<4d} k{k} s{s} up{up} | {cnt:3d} {t:8.3f} ms {100 * t / ms.sum():5.1f}% {fl / t / 1e9:7.2f}")
    L.check(ctx.lib.pp_conv_variant(-1), "variant")
    for nm in names:
        a, b = res[3][nm], res[-1][nm]
        sc = np.abs(a).max()
        print(f"  {nm}: max|exact| {sc:.4g}  max diff {np.abs(a - b).max():.3g}  rel {np.abs(a - b).max() / sc:.2e}  nan {int(np.isnan(b).sum())}")


if __name__ == "__main__":
    main()
