#!/usr/bin/env python
"""bf16-split convolution (conv_split.hip) against the exact fp32 kernel and a float64 convolution: error of both."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import _lib as L  # noqa: E402
from tests.helpers import hip_conv_op  # noqa: E402


def conv64(x, w, b, res=None, relu=0):
    n, h, ww, cin = x.shape
    cout = w.shape[0]
    xp = np.zeros((n, h + 2, ww + 2, cin), np.float64)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((n, h, ww, cout), np.float64)
    for dy in range(3):
        for dx in range(3):
            y += xp[:, dy:dy + h, dx:dx + ww] @ w[:, :, dy, dx].astype(np.float64).T
    y += b
    if relu == L.PP_RELU_FIRST:
        y = np.maximum(y, 0)
    if res is not None:
        y += res
    if relu == L.PP_RELU_LAST:
        y = np.maximum(y, 0)
    return y


def main():
    ctx = L.Context(0)
    rng = np.random.default_rng(0)
    cases = [(2, 24, 18, 64, 64), (3, 24, 18, 48, 48), (1, 40, 68, 256, 256), (2, 16, 16, 16, 32), (1, 33, 29, 128, 96),
             (2, 9, 12, 384, 384), (1, 7, 100, 32, 20), (1, 20, 34, 512, 512)]
    for n, h, w, cin, cout in cases:
        x = (rng.standard_normal((n, h, w, cin)) * np.exp(rng.standard_normal((n, h, w, cin)))).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        r = rng.standard_normal((n, h, w, cout)).astype(np.float32)
        for relu, res in ((0, None), (L.PP_RELU_LAST, r), (L.PP_RELU_FIRST, None)):
            ref = conv64(x, wt, b, res, relu)
            out = {}
            for tile in (-1, 0, 1, 2, 3):
                pass
            for v in (3, 4):
                L.check(ctx.lib.pp_conv_variant(v), "variant")
                out[v] = hip_conv_op(ctx, x, wt, b, pad=(1, 1), relu=relu, res1=res)
            L.check(ctx.lib.pp_conv_variant(-1), "variant")
            scale = np.abs(ref).max()
            e3 = np.abs(out[3] - ref).max() / scale
            e4 = np.abs(out[4] - ref).max() / scale
            r3 = np.sqrt(np.mean((out[3] - ref) ** 2)) / scale
            r4 = np.sqrt(np.mean((out[4] - ref) ** 2)) / scale
            print(f"{n}x{h}x{w} {cin}->{cout} relu{relu} res{int(res is not None)}: exact max {e3:.2e} rms {r3:.2e} | split max {e4:.2e} rms {r4:.2e}"
                  f" | nan {int(np.isnan(out[4]).sum())} | split-exact max {np.abs(out[4] - out[3]).max() / scale:.2e}", flush=True)


if __name__ == "__main__":
    main()
