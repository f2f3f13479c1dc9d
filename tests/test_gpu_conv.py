"""HIP implicit-GEMM conv vs the C oracle: bit-exact (fp32 MFMA == k-ordered fmaf chain)."""
import numpy as np
import pytest

from posepipeline_amd import _lib as L
from tests.helpers import hip_conv_op, ref_conv_op

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, k, stride, pad, dil
    (2, 16, 12, 4, 64, 3, 2, 1, 1),      # stem-like, Cin padded 3->4
    (2, 24, 18, 64, 64, 3, 1, 1, 1),
    (3, 24, 18, 48, 48, 3, 1, 1, 1),     # W48 branch 0
    (3, 12, 9, 96, 96, 3, 1, 1, 1),
    (2, 12, 9, 192, 384, 3, 2, 1, 1),
    (2, 24, 18, 256, 64, 1, 1, 0, 1),    # bottleneck 1x1
    (1, 24, 18, 64, 256, 1, 1, 0, 1),
    (2, 24, 18, 32, 17, 1, 1, 0, 1),     # head, Cout not a multiple of 4
    (4, 1, 60, 36, 1024, 3, 1, 0, 1),    # VideoPose3D expand (Cin 34->36), conv1d
    (2, 1, 90, 1024, 1024, 3, 1, 0, 9),  # dilated temporal conv
    (2, 1, 20, 1024, 51, 1, 1, 0, 1),    # shrink
    (1, 37, 53, 8, 24, 7, 2, 3, 1),      # ResNet stem shape class (7x7 s2 p3), ragged dims
    (1, 5, 7, 20, 40, 3, 1, 1, 1),       # K not a multiple of 16, tiny M
    (1, 40, 40, 16, 3, 1, 1, 0, 1),      # RPN cls-like Cout=3
    (5, 33, 29, 128, 128, 3, 1, 1, 1),   # M large enough for the PT=4 tile
    (64, 24, 18, 64, 128, 3, 1, 1, 1),   # PT=4, several channel blocks
]


@pytest.fixture(params=[0, 1, 3], ids=["two_barrier", "pipelined", "pipelined_tap_table"])
def variant(ctx, request):
    """every kernel variant (conv_igemm.hip / conv_igemm_p3.hip without and with the tap table) must give the oracle's bits"""
    L.check(ctx.lib.pp_conv_variant(request.param), "pp_conv_variant")
    yield request.param
    L.check(ctx.lib.pp_conv_variant(-1), "pp_conv_variant")
    L.check(ctx.lib.pp_conv_force(0, 0), "pp_conv_force")


@pytest.mark.parametrize("case", CASES)
def test_conv_bit_exact(ctx, case, variant):
    n, h, w, cin, cout, k, stride, pad, dil = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    kh = 1 if h == 1 else k
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, kh, k)) / np.sqrt(cin * kh * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    kw = dict(stride=stride, pad=(0 if h == 1 else pad, pad), dil=(1, dil))
    ref = ref_conv_op(x, wt, b, **kw)
    got = hip_conv_op(ctx, x, wt, b, **kw)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"max |d| = {np.abs(got - ref).max()}"


@pytest.mark.parametrize("case", [CASES[2], CASES[4], CASES[8], CASES[11], CASES[12], CASES[15]])
def test_conv_every_tile_configuration(ctx, case, variant):
    """results do not depend on the tile configuration (channel tile 16..64, pixel tile 64 / 128) of either kernel"""
    n, h, w, cin, cout, k, stride, pad, dil = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    kh = 1 if h == 1 else k
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, kh, k)) / np.sqrt(cin * kh * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    kw = dict(stride=stride, pad=(0 if h == 1 else pad, pad), dil=(1, dil))
    ref = ref_conv_op(x, wt, b, **kw)
    for ct in (1, 2, 3, 4):
        for pt in (1, 2):
            L.check(ctx.lib.pp_conv_force(ct, pt), "pp_conv_force")
            assert np.array_equal(hip_conv_op(ctx, x, wt, b, **kw), ref), (ct, pt)


def test_conv_epilogues(ctx, variant):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 12, 9, 96)).astype(np.float32)
    wt = (rng.standard_normal((48, 96, 1, 1)) / 10).astype(np.float32)
    b = rng.standard_normal(48).astype(np.float32)
    # HRNet fuse: 1x1 conv at low res, nearest-upsampled x2 / x4, accumulated onto a partial sum, identity, ReLU
    for up in (1, 2):
        r1 = rng.standard_normal((3, 12 << up, 9 << up, 48)).astype(np.float32)
        r2 = rng.standard_normal((3, 12 << up, 9 << up, 48)).astype(np.float32)
        for relu in (L.PP_RELU_NONE, L.PP_RELU_LAST):
            kw = dict(res1=r1, res2=r2, up_log2=up, relu=relu)
            assert np.array_equal(hip_conv_op(ctx, x, wt, b, **kw), ref_conv_op(x, wt, b, **kw))
    # BasicBlock tail: 3x3 + residual + ReLU
    w3 = (rng.standard_normal((96, 96, 3, 3)) / 30).astype(np.float32)
    b3 = rng.standard_normal(96).astype(np.float32)
    kw = dict(pad=(1, 1), res1=x, relu=L.PP_RELU_LAST)
    assert np.array_equal(hip_conv_op(ctx, x, w3, b3, **kw), ref_conv_op(x, w3, b3, **kw))
    # strided fuse conv with both residuals
    r1 = rng.standard_normal((3, 6, 5, 96)).astype(np.float32)
    r2 = rng.standard_normal((3, 6, 5, 96)).astype(np.float32)
    kw = dict(stride=2, pad=(1, 1), res1=r1, res2=r2, relu=L.PP_RELU_LAST)
    assert np.array_equal(hip_conv_op(ctx, x, w3, b3, **kw), ref_conv_op(x, w3, b3, **kw))
    # NCHW heatmap output, Cout = 17
    wh = (rng.standard_normal((17, 96, 1, 1)) / 10).astype(np.float32)
    bh = rng.standard_normal(17).astype(np.float32)
    assert np.array_equal(hip_conv_op(ctx, x, wh, bh, out_nchw=True), ref_conv_op(x, wh, bh, out_nchw=True))
    # VideoPose3D block: y = res[centre crop] + relu(conv)
    xt = rng.standard_normal((2, 1, 40, 64)).astype(np.float32)
    wt1 = (rng.standard_normal((64, 64, 1, 3)) / 14).astype(np.float32)
    bt = rng.standard_normal(64).astype(np.float32)
    kw = dict(dil=(1, 3), relu=L.PP_RELU_FIRST, res1=xt, res1_off_w=3)
    assert np.array_equal(hip_conv_op(ctx, xt, wt1, bt, **kw), ref_conv_op(xt, wt1, bt, **kw))
    # FPN top-down: lateral 1x1 + nearest-upsampled coarser level
    xl = rng.standard_normal((1, 10, 14, 32)).astype(np.float32)
    wl = (rng.standard_normal((16, 32, 1, 1)) / 6).astype(np.float32)
    bl = rng.standard_normal(16).astype(np.float32)
    coarse = rng.standard_normal((1, 5, 7, 16)).astype(np.float32)
    kw = dict(res1=coarse, res1_shift=1)
    assert np.array_equal(hip_conv_op(ctx, xl, wl, bl, **kw), ref_conv_op(xl, wl, bl, **kw))
