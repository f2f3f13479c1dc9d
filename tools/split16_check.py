#!/usr/bin/env python
"""The two split forms (three bf16 terms / six products; two fp16 terms / three products) and the float32-MFMA kernel against a
float64 convolution: maximum and rms error of each, on the shapes of every kernel form and on inputs of several ranges.
usage: python tools/split16_check.py [quick]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import _lib as L  # noqa: E402
from tests.helpers import hip_conv_op  # noqa: E402


def conv64(x, w, b, pad, stride=1, res=None, relu=0):
    n, h, ww, cin = x.shape
    cout, _, kh, kw = w.shape
    xp = np.zeros((n, h + 2 * pad, ww + 2 * pad, cin), np.float64)
    xp[:, pad:pad + h, pad:pad + ww] = x
    ho, wo = (h + 2 * pad - kh) // stride + 1, (ww + 2 * pad - kw) // stride + 1
    y = np.zeros((n, ho, wo, cout), np.float64)
    for dy in range(kh):
        for dx in range(kw):
            y += xp[:, dy:dy + (ho - 1) * stride + 1:stride, dx:dx + (wo - 1) * stride + 1:stride] @ w[:, :, dy, dx].astype(np.float64).T
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
    # (n, h, w, cin, cout, k, stride): tile / stream / 48-channel / 8-wave / one-tap / product / tap-gather forms
    cases = [(2, 24, 18, 64, 64, 3, 1), (3, 24, 18, 48, 48, 3, 1), (1, 40, 68, 256, 256, 3, 1), (2, 16, 16, 16, 32, 3, 1),
             (1, 33, 29, 128, 96, 3, 1), (2, 9, 12, 384, 384, 3, 1), (8, 96, 96, 256, 256, 3, 1), (2, 12, 20, 1024, 256, 1, 1),
             (2, 12, 20, 1024, 96, 1, 2), (2, 40, 68, 256, 1024, 1, 1), (2, 40, 68, 128, 128, 3, 2)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        cases = cases[:3]
    dists = {
        "normal": lambda s: rng.standard_normal(s),
        "heavy": lambda s: rng.standard_normal(s) * np.exp(2 * rng.standard_normal(s)),          # ~6 orders of magnitude
        "relu": lambda s: np.maximum(rng.standard_normal(s), 0) * 3,                               # same-sign data
        "tiny": lambda s: rng.standard_normal(s) * 1e-6,
        "large": lambda s: rng.standard_normal(s) * 3e3,
        "1e-30": lambda s: rng.standard_normal(s) * 1e-30,
        "1e+7": lambda s: rng.standard_normal(s) * 1e7,                                            # above the float16 range
        # every sample of the batch at its own magnitude (1e-6 .. 1e4): the activation scale is per sample
        "persample": lambda s: rng.standard_normal(s) * (10.0 ** np.linspace(-6, 4, s[0])).reshape(-1, 1, 1, 1),
    }
    print("case | dist | exact max rms | bf16x3 max rms | f16x2 max rms | f16/exact rms ratio")
    worst = 0.0
    for (n, h, w, cin, cout, k, stride) in cases:
        for dname, dist in dists.items():
            x = dist((n, h, w, cin)).astype(np.float32)
            wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
            if dname == "relu":
                wt = np.abs(wt)
            # per-channel weight ranges over 8 orders of magnitude (BN-folded weights)
            wt *= np.exp(3 * rng.standard_normal((cout, 1, 1, 1))).astype(np.float32)
            b = (rng.standard_normal(cout) * np.abs(x).mean()).astype(np.float32)
            pad = k // 2
            ref = conv64(x, wt, b, pad, stride, None, 0)
            out = {}
            for mode in ("exact", "split_bf16", "split_f16"):
                with L.default_numerics(mode):
                    out[mode] = hip_conv_op(ctx, x, wt, b, stride=stride, pad=(pad, pad), relu=0)
            # per-channel scale (the channels' ranges differ by orders of magnitude)
            # (and per sample: "persample" spreads the batch over ten orders)
            scale = np.abs(ref).reshape(n, -1, cout).max(1).reshape(n, 1, 1, cout) + 1e-300
            st = []
            for mode in ("exact", "split_bf16", "split_f16"):
                d = (out[mode] - ref) / scale
                st += [np.abs(d).max(), np.sqrt(np.mean(d * d))]
            ratio = st[5] / max(st[1], 1e-30)
            worst = max(worst, ratio)
            same = np.array_equal(out["split_f16"], out["exact"])
            print(f"{n}x{h}x{w} {cin}->{cout} k{k}s{stride} | {dname:9s} | {st[0]:.2e} {st[1]:.2e} | {st[2]:.2e} {st[3]:.2e} | {st[4]:.2e} {st[5]:.2e} | {ratio:.2f}"
                  f"{' (f16 == exact: kernel did not run?)' if same else ''}{' NaN' if not np.isfinite(out['split_f16']).all() else ''}", flush=True)
    print(f"worst f16 / exact rms ratio: {worst:.2f}")


if __name__ == "__main__":
    main()
