"""ViTPose path (BASELINE.json configs[4], bf16 MFMA): HIP kernels vs the CPU oracle (oracle/vit.py).

Tolerances.  The bf16 MFMA accumulates in fp32 in an order the ISA does not architect, so these kernels are compared
with a float64 evaluation of the SAME bf16-rounded operands: fp32 outputs within 2e-4 relative (K <= 5120 terms),
bf16 outputs within one bf16 ulp (2^-8 relative).  The whole network (oracle with bf16 rounding at the same points)
agrees to a few 1e-3 of the heatmap range; the decoded keypoints tolerance is stated in test_gpu_vit_topdown.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import vit as OV
from posepipeline_amd import _lib as L
from posepipeline_amd.models import vitpose as MV
from posepipeline_amd.program import Net, ProgramBuilder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = L.Context(0)
    yield c
    c.close()


class Dev:
    """numpy array <-> device buffer"""

    def __init__(self, ctx, arr=None, nbytes=None):
        self.ctx = ctx
        self.nbytes = int(arr.nbytes if arr is not None else nbytes)
        self.ptr = ctx.malloc(self.nbytes)
        if arr is not None:
            ctx.h2d(self.ptr, np.ascontiguousarray(arr))

    def get(self, shape, dtype):
        out = np.empty(shape, dtype)
        self.ctx.d2h(out, self.ptr)
        return out

    def free(self):
        self.ctx.free(self.ptr)


def _close_bf16(got_bits, ref_f32, ulps=1.0):
    got = OV.bf16_from_bits(got_bits).astype(np.float64)
    ref = ref_f32.astype(np.float64)
    tol = ulps * 2.0 ** -8 * np.abs(ref) + 1e-5
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - ref).max()))


@pytest.mark.parametrize("m,n,k,act,use_res,res_mod,out_bf16", [
    (256, 128, 64, 0, False, 0, 0),
    (300, 256, 192, 0, True, 0, 0),          # ragged M (tail rows masked)
    (384, 384, 1280, 1, False, 0, 1),        # GELU, bf16 out
    (576, 1280, 640, 0, True, 192, 0),       # residual broadcast over row % 192 (position embedding form)
    (1100, 640, 2560, 0, True, 0, 0),        # in-place residual stream form, ragged 256-row tiles
    (256, 256, 128, 0, False, 0, 0),         # N % 256 == 0 and K % 128 == 0: the shapes the ping-pong form (10, the default there) takes; 2 K tiles
    (300, 512, 256, 0, True, 0, 0),          # ... ragged M
    (640, 512, 1280, 1, False, 0, 1),        # ... GELU, bf16 out
    (1100, 768, 2560, 0, True, 0, 0),        # ... 40 K tiles, residual
])
@pytest.mark.parametrize("cfg", ["", "0", "1", "2", "10"])     # the default selection and every tile configuration of gemm_bf16.hip
def test_gemm_bf16(ctx, monkeypatch, cfg, m, n, k, act, use_res, res_mod, out_bf16):
    if cfg:
        monkeypatch.setenv("POSEPIPE_GEMM_CFG", cfg)
    else:
        monkeypatch.delenv("POSEPIPE_GEMM_CFG", raising=False)
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k), dtype=np.float32)
    w = rng.standard_normal((n, k), dtype=np.float32) / np.float32(np.sqrt(k))
    bias = rng.standard_normal(n, dtype=np.float32)
    res = rng.standard_normal((res_mod if res_mod else m, n), dtype=np.float32) if use_res else None
    ref = OV.linear(a, w, bias, True)
    if act:
        ref = OV.gelu(ref)
    if use_res:
        ref = (ref + (np.tile(res, (m // res_mod, 1)) if res_mod else res)).astype(np.float32)
    da, dw, db = Dev(ctx, OV.bf16_bits(a)), Dev(ctx, OV.bf16_bits(w)), Dev(ctx, bias)
    dres = Dev(ctx, res) if use_res else None
    dc = Dev(ctx, nbytes=m * n * (2 if out_bf16 else 4))
    L.check(ctx.lib.pp_gemm_bf16(ctx.handle, da.ptr, dw.ptr, db.ptr, dres.ptr if dres else None, res_mod, dc.ptr, m, n, k,
                                 act, out_bf16), "pp_gemm_bf16")
    ctx.synchronize()
    if out_bf16:
        _close_bf16(dc.get((m, n), np.uint16), ref)
    else:
        got = dc.get((m, n), np.float32)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    for d in (da, dw, db, dc) + ((dres,) if dres else ()):
        d.free()


def test_gemm_bf16_identity_asymmetric(ctx):
    """A = I against an asymmetric W catches a transposed / permuted fragment mapping exactly (no rounding involved)."""
    n = k = 256
    w = (np.arange(n)[:, None] * 3 + np.arange(k)[None, :] * 7) % 251
    w = w.astype(np.float32)                                # integers < 256: exact in bf16
    a = np.eye(k, dtype=np.float32)
    da, dw, dc = Dev(ctx, OV.bf16_bits(a)), Dev(ctx, OV.bf16_bits(w)), Dev(ctx, nbytes=k * n * 4)
    L.check(ctx.lib.pp_gemm_bf16(ctx.handle, da.ptr, dw.ptr, None, None, 0, dc.ptr, k, n, k, 0, 0), "pp_gemm_bf16")
    ctx.synchronize()
    assert np.array_equal(dc.get((k, n), np.float32), w.T)
    for d in (da, dw, dc):
        d.free()


def test_gemm_bf16_rejects_bad_shapes(ctx):
    d = Dev(ctx, nbytes=1 << 20)
    assert ctx.lib.pp_gemm_bf16(ctx.handle, d.ptr, d.ptr, None, None, 0, d.ptr, 128, 100, 64, 0, 0) == -1   # n % 128
    assert ctx.lib.pp_gemm_bf16(ctx.handle, d.ptr, d.ptr, None, None, 0, d.ptr, 128, 128, 80, 0, 0) == -1    # k % 64
    assert ctx.lib.pp_attention_bf16(ctx.handle, d.ptr, 1, 100, 4, 80, d.ptr) == -4                          # not built
    d.free()


@pytest.mark.parametrize("rows,dim", [(7, 1280), (192, 640), (33, 768), (5, 2048), (9, 100)])
def test_layernorm(ctx, rows, dim):
    rng = np.random.default_rng(rows * dim)
    x = (rng.standard_normal((rows, dim), dtype=np.float32) * 3 + 1).astype(np.float32)
    g = rng.uniform(0.5, 1.5, dim).astype(np.float32)
    b = rng.standard_normal(dim, dtype=np.float32)
    ref = OV.layernorm(x, g, b)
    dx, dg, db = Dev(ctx, x), Dev(ctx, g), Dev(ctx, b)
    dy = Dev(ctx, nbytes=rows * dim * 4)
    L.check(ctx.lib.pp_layernorm(ctx.handle, dx.ptr, dg.ptr, db.ptr, rows, dim, OV.LN_EPS, dy.ptr, 0), "pp_layernorm")
    ctx.synchronize()
    np.testing.assert_allclose(dy.get((rows, dim), np.float32), ref, rtol=1e-5, atol=2e-5)
    L.check(ctx.lib.pp_layernorm(ctx.handle, dx.ptr, dg.ptr, db.ptr, rows, dim, OV.LN_EPS, dy.ptr, 1), "pp_layernorm")
    ctx.synchronize()
    _close_bf16(dy.get((rows, dim), np.uint16), ref)
    for d in (dx, dg, db, dy):
        d.free()


@pytest.mark.parametrize("heads,hd", [(8, 80), (4, 64)])
def test_attention(ctx, heads, hd):
    batch, t, dim = 3, 192, heads * hd
    rng = np.random.default_rng(hd)
    qkv = rng.standard_normal((batch * t, 3 * dim), dtype=np.float32)
    qkv[:, :dim] *= 2.0                                       # peaky softmax rows as well as flat ones
    ref = OV.attention(qkv, batch, t, heads, True)
    dq = Dev(ctx, OV.bf16_bits(qkv))
    do = Dev(ctx, nbytes=batch * t * dim * 2)
    L.check(ctx.lib.pp_attention_bf16(ctx.handle, dq.ptr, batch, t, heads, hd, do.ptr), "pp_attention_bf16")
    ctx.synchronize()
    got = OV.bf16_from_bits(do.get((batch * t, dim), np.uint16))
    # numerators are bf16 (2^-9 relative each, averaged over 192 keys) and the output is rounded to bf16 once more
    np.testing.assert_allclose(got, ref, rtol=2.0 ** -7, atol=4e-3)
    assert np.abs(got - ref).mean() < 1e-3
    dq.free(); do.free()


def test_deconv_as_four_convs(ctx):
    """ConvTranspose2d(4, 2, 1) == four 2x2 convolutions + depth_to_space (program.deconv4x4s2) against the direct
    float64 definition."""
    rng = np.random.default_rng(5)
    cin, cout, h, w, n = 32, 16, 5, 7, 3
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = rng.standard_normal((cin, cout, 4, 4), dtype=np.float32) * np.float32(0.2)
    bias = rng.standard_normal(cout, dtype=np.float32)
    b = ProgramBuilder()
    xin = b.buf(h, w, cin, name="input")
    y = b.deconv4x4s2(xin, wt, bias)
    out = b.buf(2 * h, 2 * w, cout, name="output")
    b.conv(y, np.eye(cout, dtype=np.float32).reshape(cout, cout, 1, 1), None, out=out, name="copy")
    net = Net(ctx, b.build(), n)
    got = net.forward(x)
    ref = OV.conv_transpose_4s2p1(x, wt) + bias.astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    net.close()


SMALL = MV.VitPoseSpec(dim=640, depth=2, heads=8, mlp_ratio=4, num_joints=17, deconv=(64, 64))


def test_head_major_qkv_layout_is_bit_identical(ctx, monkeypatch):
    """POSEPIPE_VIT_HEAD_MAJOR=1 (qkv GEMM writes [3][sample][head][token][d], attention reads contiguous slabs) is the same
    arithmetic on the same values: identical heat-maps."""
    p = MV.synth_params(SMALL, seed=8)
    prog = MV.build_vitpose_program(SMALL, p)
    x = np.zeros((2, SMALL.in_h, SMALL.in_w, 4), np.float32)
    x[..., :3] = np.random.default_rng(12).standard_normal((2, SMALL.in_h, SMALL.in_w, 3), dtype=np.float32)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("POSEPIPE_VIT_HEAD_MAJOR", flag)      # read when the encoder is created
        net = Net(ctx, prog, 2)
        outs.append(net.forward(x))
        net.close()
    assert np.array_equal(outs[0], outs[1])


def test_vitpose_small_program(ctx):
    """patch embedding + 2-block encoder + head, program vs oracle (bf16 rounding emulated at the same points)."""
    p = MV.synth_params(SMALL, seed=3)
    prog = MV.build_vitpose_program(SMALL, p)
    n = 3
    rng = np.random.default_rng(11)
    x = np.zeros((n, SMALL.in_h, SMALL.in_w, 4), np.float32)
    x[..., :3] = rng.standard_normal((n, SMALL.in_h, SMALL.in_w, 3), dtype=np.float32)
    net = Net(ctx, prog, n)
    got = net.forward(x)                                                     # [n][K][64][48] (NCHW in an NHWC-shaped array)
    got = got.reshape(n, SMALL.num_joints, *SMALL.heatmap_hw)
    ref = OV.forward(x, p, SMALL, emulate_bf16=True)
    ref32 = OV.forward(x, p, SMALL, emulate_bf16=False)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max()) / scale
    err32 = float(np.abs(got - ref32).max()) / scale
    print(f"vitpose small: max|hip - oracle_bf16| = {err:.2e} of range, vs fp32 model {err32:.2e}")
    assert err < 5e-3, err            # fp32 accumulation order + occasional 1-ulp bf16 flips downstream
    assert err32 < 5e-2, err32        # the price of bf16 operands against the fp32 model
    # patch embedding alone is an fp32 convolution: exact against its float64 statement to fp32 rounding
    tok, _ = OV.patch_embed(x, p["backbone.patch_embed.proj.weight"], p["backbone.patch_embed.proj.bias"])
    net.run(n, 0, 1)
    ctx.synchronize()
    got_tok = net.read(prog.ops[0].out, n).reshape(n, SMALL.tokens, SMALL.dim)
    np.testing.assert_allclose(got_tok, tok, rtol=1e-4, atol=1e-4)
    net.close()


def test_deconv_bf16_op(ctx):
    """PP_OP_DECONV_BF16 (one GEMM over the 16 kernel taps + gather) against the direct float64 ConvTranspose2d(4, 2, 1)
    of the same bf16-rounded operands: fp32 accumulation tolerance only; borders (absent taps) included."""
    rng = np.random.default_rng(15)
    cin, cout, h, w, n = 128, 24, 5, 7, 3
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = rng.standard_normal((cin, cout, 4, 4), dtype=np.float32) * np.float32(0.1)
    bias = rng.standard_normal(cout, dtype=np.float32)
    b = ProgramBuilder()
    xin = b.buf(h, w, cin, name="input")
    y = b.deconv4x4s2_bf16(xin, wt, bias, relu=L.PP_RELU_LAST)
    out = b.buf(2 * h, 2 * w, cout, name="output")
    b.conv(y, np.eye(cout, dtype=np.float32).reshape(cout, cout, 1, 1), None, out=out, name="copy")
    net = Net(ctx, b.build(), n)
    got = net.forward(x)
    ref = np.maximum(OV.conv_transpose_4s2p1(OV.bf16_round(x), OV.bf16_round(wt)) + bias.astype(np.float64), 0.0)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    assert (got == 0).any() and (got > 0).any()
    net.close()


@pytest.mark.parametrize("cfg", ["2", "10"])
def test_two_stage_schedules_are_race_free_and_bit_equal_to_single_stage(ctx, monkeypatch, cfg):
    """The pipelined schedules (loads of K tile k + 1 in flight while tile k is multiplied; one barrier per K step) on a
    chip-filling problem, 25 launches: every launch must reproduce the first bit for bit, and all of them the plain
    single-stage kernel (every output element is the same MFMA chain over K whatever the tiling), so a landing-order race
    on an LDS stage cannot hide behind a tolerance."""
    m, n, k = 6144, 3840, 1280
    rng = np.random.default_rng(99)
    a = OV.bf16_bits(rng.standard_normal((m, k), dtype=np.float32))
    w = OV.bf16_bits(rng.standard_normal((n, k), dtype=np.float32) / np.float32(np.sqrt(k)))
    da, dw, dc = Dev(ctx, a), Dev(ctx, w), Dev(ctx, nbytes=m * n * 4)

    def run(c):
        monkeypatch.setenv("POSEPIPE_GEMM_CFG", c)
        L.check(ctx.lib.pp_gemm_bf16(ctx.handle, da.ptr, dw.ptr, None, None, 0, dc.ptr, m, n, k, 0, 0), "pp_gemm_bf16")
        ctx.synchronize()
        return dc.get((m, n), np.float32)

    base = run("0")
    if int(cfg) >= 10:
        # the ping-pong form multiplies with v_mfma_f32_32x32x16_bf16: 16 k per instruction instead of 32, another summation
        # order -- float32-close to the single-stage kernel, and bit-equal to ITSELF over 50 launches
        first = run(cfg)
        np.testing.assert_allclose(first, base, rtol=0, atol=2e-5 * float(np.abs(base).max()))
        assert not np.array_equal(first, np.zeros_like(first))
        base = first
    for _ in range(50 if int(cfg) >= 10 else 25):
        assert np.array_equal(run(cfg), base)
    for d in (da, dw, dc):
        d.free()
