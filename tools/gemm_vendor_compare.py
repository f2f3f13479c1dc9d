"""Where does the vendor library sit on the ViT-H GEMM shapes?   python tools/gemm_vendor_compare.py [batch]

Measurement aid, not product code: times `torch.nn.functional.linear` (hipBLASLt / rocBLAS behind PyTorch-ROCm; PLAIN bf16 GEMM +
bias, no GELU / residual / head-major store) and this repository's `pp_gemm_bf16` (fused epilogues) on the same box, same shapes
(M = batch * 192 tokens), so that the ">= 0.50 of 2.5 PFLOP/s" question has the library's own number beside it.
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from posepipeline_amd import _lib as L  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    m, d = batch * 192, 1280
    ctx = L.Context(0)
    dev = torch.device("cuda:0")
    print(f"M = {m} tokens; TFLOP/s (us per launch)")
    for name, n, k, act, obf, res in (("qkv", 3 * d, d, 0, 1, 0), ("proj", d, d, 0, 0, 1), ("fc1", 4 * d, d, 1, 1, 0), ("fc2", d, 4 * d, 0, 0, 1)):
        fl = 2.0 * m * n * k
        # vendor: plain GEMM + bias, bf16 in / bf16 out
        a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        b = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            torch.nn.functional.linear(a, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.nn.functional.linear(a, w, b)
        e1.record()
        torch.cuda.synchronize()
        ms_v = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20):
            torch.matmul(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        ms_p = e0.elapsed_time(e1) / 20
        del a, w, b
        # own kernel with its fused epilogue
        rng = np.random.default_rng(1)

        def dev_bf16(cnt):
            bits = (rng.standard_normal(cnt, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
            p = ctx.malloc(bits.nbytes)
            ctx.h2d(p, bits)
            return p
        pa, pw = dev_bf16(m * k), dev_bf16(n * k)
        bias = ctx.malloc(n * 4)
        ctx.h2d(bias, np.zeros(n, np.float32))
        c = ctx.malloc(m * n * 4)
        ctx.h2d(c, np.zeros(m * n, np.float32))

        def own():
            L.check(ctx.lib.pp_gemm_bf16(ctx.handle, pa, pw, bias, c if res else None, 0, c, m, n, k, act, obf))
        own()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(20):
            own()
        ms_o = ctx.timer_stop() / 20
        for p in (pa, pw, bias, c):
            ctx.free(p)
        print(f"{name:5s} N={n:5d} K={k:5d}: vendor linear+bias {fl / ms_v / 1e9:7.1f} ({ms_v * 1e3:6.1f})   vendor matmul {fl / ms_p / 1e9:7.1f} ({ms_p * 1e3:6.1f})"
              f"   own, fused epilogue {fl / ms_o / 1e9:7.1f} ({ms_o * 1e3:6.1f})", flush=True)


if __name__ == "__main__":
    main()
