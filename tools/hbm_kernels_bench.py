#!/usr/bin/env python
"""The HBM-bound kernels of the path in isolation, device-resident inputs, HIP-event timing: algorithmic bytes / time against the
8 TB/s HBM3E peak (and the ~6.3 TB/s a float4 copy sustains on this part).

  flip_merge_decode   64 persons x 17 joints x 96 x 72 (HRNet-W48 384x288): 2 heat-map reads per joint, 12 B written
  nv12_to_bgr         64 frames of 1080p NV12 (3.1 MB in, 6.2 MB out per frame)
  det_preprocess      64 frames of 1080p BGR -> 640 x 1088 x 4 float32 (resize + normalise + pad)
  crop_affine         64 persons cropped from 1080p frames to 2 x 384 x 288 x 4 float32 (both flips)

usage: python tools/hbm_kernels_bench.py [decode|nv12|all] [--generic]   (put under tools/pmc_kernel.sh for FETCH / WRITE counters)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posepipeline_amd import _lib as L  # noqa: E402
from posepipeline_amd.models import hrnet  # noqa: E402


def timed(ctx, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ts.append(ctx.timer_stop())
    return float(np.median(ts)), float(np.min(ts))


def bench_decode(ctx, n=64, k=17, h=96, w=72, post=1, ks=17):
    rng = np.random.default_rng(0)
    hm = rng.standard_normal((n, k, h, w)).astype(np.float32)
    hf = rng.standard_normal((n, k, h, w)).astype(np.float32)
    cs = np.tile(np.array([[960.0, 540.0, 1.9, 2.53]], np.float32), (n, 1))
    d_hm, d_hf, d_cs, d_kp = (ctx.malloc(a.nbytes) for a in (hm, hf, cs, np.zeros((n, k, 3), np.float32)))
    ctx.h2d(d_hm, hm)
    ctx.h2d(d_hf, hf)
    ctx.h2d(d_cs, cs)
    perm = np.ascontiguousarray(hrnet.flip_perm(k), np.int32)

    def run():
        L.check(ctx.lib.pp_flip_merge_decode(ctx.handle, C.c_void_p(d_hm), C.c_void_p(d_hf), n, k, h, w, L.ptr(perm), 1, post, ks,
                                             C.c_void_p(d_cs), C.c_void_p(d_kp), None, L.PP_MEM_DEVICE), "decode")
    med, best = timed(ctx, run)
    nbytes = 2 * hm.nbytes + n * k * 12
    print(f"flip_merge_decode {n} x {k} x {h}x{w} post {post} kernel {ks}: {med * 1e3:8.1f} us (best {best * 1e3:.1f})  {nbytes / 1e6:.1f} MB  "
          f"{nbytes / med / 1e9:.3f} TB/s algorithmic = {nbytes / med / 1e9 / 8.0:.3f} of 8 TB/s   [includes the per-call H2D of the flip "
          f"permutation and the stream synchronisation of the C entry point]")


def bench_nv12(ctx, n=64, h=1080, w=1920):
    rng = np.random.default_rng(1)
    nv = rng.integers(0, 256, (n, h * 3 // 2, w), dtype=np.uint8)
    d_in, d_out = ctx.malloc(nv.nbytes), ctx.malloc(n * h * w * 3)
    ctx.h2d(d_in, nv)

    def run():
        L.check(ctx.lib.pp_nv12_to_bgr(ctx.handle, C.c_void_p(d_in), n, h, w, C.c_void_p(d_out)), "nv12")
    med, best = timed(ctx, run)
    nbytes = nv.nbytes + n * h * w * 3
    print(f"nv12_to_bgr {n} x {h}x{w}: {med * 1e3:8.1f} us (best {best * 1e3:.1f})  {nbytes / 1e6:.1f} MB  {nbytes / med / 1e9:.3f} TB/s algorithmic = "
          f"{nbytes / med / 1e9 / 8.0:.3f} of 8 TB/s")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "all"
    ctx = L.Context(0)
    if what in ("decode", "all"):
        if os.environ.get("DECODE_ONE"):
            bench_decode(ctx, n=int(os.environ["DECODE_ONE"]))
            return
        bench_decode(ctx)                      # configs[2] / [3]: W48 384x288, DARK
        bench_decode(ctx, h=64, w=48, post=0)  # configs[1]: W32 256x192, 'default'
        bench_decode(ctx, h=64, w=48, post=2, ks=11)   # ViTPose: UDP
        bench_decode(ctx, n=256)               # 4 persons per frame
    if what in ("nv12", "all"):
        bench_nv12(ctx)
        if not os.environ.get("NV12_ONE"):
            bench_nv12(ctx, h=480, w=854)      # w % 4 == 2: the 16-bit path


if __name__ == "__main__":
    main()
