"""Micro-benchmarks of the ViT encoder kernels (bf16 MFMA path): python tools/vit_bench.py [gemm|layer|all] [batch]

gemm : the four GEMM shapes of a ViT-H block at M = batch * 192 tokens, TFLOP/s each (HIP events, 20 launches)
layer: LayerNorm, attention and one full ViT-H encoder (depth from argv[3], default 32)
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from posepipeline_amd import _lib as L  # noqa: E402


def dev_rand_bf16(ctx, n, seed):
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal(n, dtype=np.float32))
    bits = (a.view(np.uint32) >> 16).astype(np.uint16)
    p = ctx.malloc(bits.nbytes)
    ctx.h2d(p, bits)
    return p


def time_call(ctx, fn, reps=20):
    fn()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


def bench_gemm(ctx, batch):
    m = batch * 192
    d = 1280
    for name, n, k, act, obf, res in (("qkv", 3 * d, d, 0, 1, 0), ("proj", d, d, 0, 0, 1), ("fc1", 4 * d, d, 1, 1, 0),
                                      ("fc2", d, 4 * d, 0, 0, 1)):
        a = dev_rand_bf16(ctx, m * k, 1)
        w = dev_rand_bf16(ctx, n * k, 2)
        bias = ctx.malloc(n * 4)
        ctx.h2d(bias, np.zeros(n, np.float32))
        c = ctx.malloc(m * n * 4)
        ctx.h2d(c, np.zeros(m * n, np.float32))
        ms = time_call(ctx, lambda: L.check(ctx.lib.pp_gemm_bf16(ctx.handle, a, w, bias, c if res else None, 0, c, m, n, k,
                                                                 act, obf)))
        print(f"gemm {name:5s} M={m} N={n} K={k}: {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s", flush=True)
        for p in (a, w, bias, c):
            ctx.free(p)


def bench_layer(ctx, batch, depth):
    from posepipeline_amd.models import vitpose as MV
    from posepipeline_amd.program import Net
    m, d, heads = batch * 192, 1280, 16
    x = ctx.malloc(m * d * 4)
    ctx.h2d(x, np.random.default_rng(0).standard_normal(m * d, dtype=np.float32))
    g = ctx.malloc(d * 4)
    ctx.h2d(g, np.ones(d, np.float32))
    y = ctx.malloc(m * d * 4)
    ms = time_call(ctx, lambda: L.check(ctx.lib.pp_layernorm(ctx.handle, x, g, g, m, d, 1e-6, y, 1)))
    print(f"layernorm {m}x{d}: {ms * 1e3:8.1f} us  {m * d * 6 / ms / 1e6:7.1f} GB/s", flush=True)
    qkv = dev_rand_bf16(ctx, m * 3 * d, 3)
    ms = time_call(ctx, lambda: L.check(ctx.lib.pp_attention_bf16(ctx.handle, qkv, batch, 192, heads, 80, y)))
    print(f"attention b={batch}: {ms * 1e3:8.1f} us  {4.0 * batch * heads * 192 * 192 * 80 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    spec = MV.VitPoseSpec(depth=depth)
    t0 = time.time()
    p = MV.synth_params(spec, 0)
    prog = MV.build_vitpose_program(spec, p)
    del p
    print(f"program built in {time.time() - t0:.1f} s, {prog.flops / 1e9:.1f} GFLOP/sample", flush=True)
    net = Net(ctx, prog, batch)
    net.set_lanes(False)
    ms_ops = net.profile(batch)
    ms_ops = net.profile(batch)
    for name, ms_, fl in zip(prog.op_names, ms_ops, prog.op_flops):
        print(f"  {name:24s} {ms_:8.3f} ms  {fl * batch / ms_ / 1e9 if ms_ > 0 else 0:8.1f} TFLOP/s")
    tot = float(ms_ops.sum())
    print(f"vitpose depth {depth} batch {batch}: {tot:.2f} ms = {batch / tot * 1e3:.0f} passes/s, "
          f"{prog.flops * batch / tot / 1e9:.1f} TFLOP/s effective", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    ctx = L.Context(0)
    if what in ("gemm", "all"):
        bench_gemm(ctx, batch)
    if what in ("layer", "all"):
        bench_layer(ctx, batch, depth)
