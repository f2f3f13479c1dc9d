#!/usr/bin/env python
"""One convolution layer at a sweep of batch sizes and tile configurations: TFLOP/s vs how the grid quantises onto the chip.

  python tools/conv_probe.py H W Cin Cout K [stride] --batches 19,38,57,64,76,128 [--res] [--cfg 0,0 3,2 3,1]

blocks = ceil(batch * Hout * Wout / (64 PT)) * ceil(Cout / (16 CT)); the chip holds 256 CUs x 4 resident blocks (PT = 2:
occupancy by registers).  Separates steady-state efficiency (many rounds) from round quantisation (blocks / 1024)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("POSEPIPE_CONV_TUNING", "0")
from posepipeline_amd import _lib as L  # noqa: E402
from posepipeline_amd.program import Net, ProgramBuilder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dims", type=int, nargs="+", help="H W Cin Cout K [stride]")
    ap.add_argument("--batches", default="16,32,64,128")
    ap.add_argument("--cfg", nargs="*", default=["0,0"])
    ap.add_argument("--res", action="store_true", help="residual add + ReLU epilogue (BasicBlock conv2)")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--variants", default="0", help="kernel variants to time: 0 two-barrier, 1 three-stage pipelined")
    a = ap.parse_args()
    h, w, cin, cout, k = a.dims[:5]
    stride = a.dims[5] if len(a.dims) > 5 else 1
    batches = [int(b) for b in a.batches.split(",")]
    rng = np.random.default_rng(0)
    pb = ProgramBuilder()
    x = pb.buf(h, w, cin, name="input")
    wt = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    wo = (w + 2 * (k // 2) - k) // stride + 1
    r = pb.buf(ho, wo, cout, name="res") if a.res else -1
    pb.conv(x, wt, np.zeros(cout, np.float32), stride=stride, pad=k // 2, relu=L.PP_RELU_LAST, res1=r, name="probe")
    prog = pb.build()
    ctx = L.Context(0)
    net = Net(ctx, prog, max_batch=max(batches))
    # real activations: all-zero buffers run ~15 % faster (the chip clocks to its power budget) and flatter every kernel
    net.forward(rng.standard_normal((max(batches), h, w, cin)).astype(np.float32), out_name="input")
    flop = 2.0 * ho * wo * cout * cin * k * k
    print(f"{h}x{w} {cin}->{cout} k{k} s{stride}{' +res' if a.res else ''}: {flop / 1e9:.3f} GFLOP per sample")
    for variant, cfg in [(int(v), c) for v in a.variants.split(",") for c in a.cfg]:
        ct, pt = (int(v) for v in cfg.split(","))
        L.check(ctx.lib.pp_conv_force(ct, pt), "pp_conv_force")
        L.check(ctx.lib.pp_conv_variant(variant), "pp_conv_variant")
        for b in batches:
            net.profile(b)
            ms = float(np.median([net.profile(b)[0] for _ in range(a.reps)]))
            m = b * ho * wo
            ect, ept = (ct or "auto"), (pt or "auto")
            blocks = ""
            if ct and pt:
                nb = -(-m // (64 * pt)) * -(-((cout + 15) // 16) // ct)
                blocks = f" blocks {nb:6d} = {nb / 1024:5.2f} rounds"
            print(f"  v{variant} cfg {ect},{ept} batch {b:4d}: {ms * 1e3:9.1f} us  {flop * b / ms / 1e9:7.2f} TFLOP/s{blocks}")
    L.check(ctx.lib.pp_conv_force(0, 0), "pp_conv_force")
    L.check(ctx.lib.pp_conv_variant(-1), "pp_conv_variant")


if __name__ == "__main__":
    main()
