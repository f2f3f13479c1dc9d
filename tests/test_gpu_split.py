"""The library's default convolution numerics: float32 convolutions on the bf16 / fp16 matrix cores (conv_split.hip).

Two forms.  bf16: every float32 operand is split exactly into three bfloat16 values and six of the nine partial products are
accumulated in float32; the dropped terms are <= 2^-23 of a product.  fp16 (round 5, the default): two float16 terms per operand
(22 significand bits under per-channel / per-sample power-of-two scales), three products.  The result is NOT bit-identical to oracle/conv_ref.c (the other GPU
tests pin the float32-MFMA kernels to it bit for bit), so this file states what the default path guarantees instead:
  * per layer, the error against a float64 convolution is that of the float32 FMA chain (not larger than 1.25x);
  * whole networks agree with the bit-exact kernels to ~3e-6 of the output range;
  * end to end, track ids / frame indices are identical and 2D / 3D joints are within north_star's 1e-3 px / mm
    (measured: ~1e-5) of the bit-exact path, which equals the oracle.
"""
import numpy as np
import pytest

from posepipeline_amd import _lib as L
from posepipeline_amd import ops
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.models import videopose3d as vp3d
from posepipeline_amd.program import Net, ProgramBuilder
from tests.helpers import hip_conv_op
from tests.test_gpu_detector import synth_frame
from tests.test_gpu_pipeline import synth_clip

pytestmark = pytest.mark.gpu

TOL_PX = 1e-3        # north_star: 2D joints within 1e-3 px
TOL_MM = 1e-3        # 3D joints within 1e-3 mm (VideoPose3D works in metres: 1e-6)


@pytest.fixture(params=["f16", "bf16"])
def lib(ctx, request):
    """the tests below switch the process-wide DEFAULT numerics (pp_conv_exact) around the creation of their nets / their
    single-op calls; a net keeps what it was created with.  Every test runs on BOTH split forms (pp_conv_split_kind): two float16
    terms / three products (the default since round 5) and three bfloat16 terms / six products"""
    L.check(ctx.lib.pp_conv_split_kind(1 if request.param == "f16" else 0), "pp_conv_split_kind")
    yield ctx.lib
    L.check(ctx.lib.pp_conv_split_kind(-1), "pp_conv_split_kind")
    L.check(ctx.lib.pp_conv_exact(1), "pp_conv_exact")


def both(lib, fn):
    """fn() under the bit-exact kernels and under the default (split) kernels"""
    out = []
    for exact in (1, 0):
        L.check(lib.pp_conv_exact(exact), "pp_conv_exact")
        out.append(fn())
    L.check(lib.pp_conv_exact(1), "pp_conv_exact")
    return out


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


def check_layer(lib, ctx, x, wt, b, pad, stride=1, res=None, relu=0):
    ref = conv64(x, wt, b, pad, stride, res, relu)
    exact, split = both(lib, lambda: hip_conv_op(ctx, x, wt, b, stride=stride, pad=(pad, pad), relu=relu, res1=res))
    assert np.isfinite(split).all() and split.shape == ref.shape
    scale = np.abs(ref).max()
    e_exact, e_split = np.abs(exact - ref).max() / scale, np.abs(split - ref).max() / scale
    r_exact, r_split = np.sqrt(np.mean((exact - ref) ** 2)) / scale, np.sqrt(np.mean((split - ref) ** 2)) / scale
    assert not np.array_equal(exact, split), "the split kernel did not run (results bit-identical to the fp32 kernel)"
    # rms: the robust statistic (measured 0.65 - 0.86 of the float32 chain's on every form, tools/c48_err.py); the maximum over
    # ~1e5 outputs of heavy-tailed inputs fluctuates from case to case (measured ratios 0.4 - 1.44)
    assert e_split <= 1.5 * e_exact + 1e-7, (e_split, e_exact)
    assert r_split <= 1.1 * r_exact + 1e-8, (r_split, r_exact)
    assert np.abs(split - exact).max() <= 1e-5 * scale
    return e_split


# n, h, w, cin, cout: every tile shape of the 3x3 kernel (8x32, 4x64, 16x16, 32x8 pixel tiles), ragged maps, channel counts
# that are not multiples of 32, 1..24 channel chunks
CASES_3X3 = [(1, 8, 32, 32, 64), (2, 4, 64, 16, 32), (2, 16, 16, 48, 48), (1, 32, 8, 64, 96), (1, 20, 34, 256, 256),
             (3, 24, 18, 96, 96), (1, 33, 29, 128, 100), (2, 9, 12, 384, 384), (1, 7, 100, 32, 20), (1, 40, 68, 64, 192),
             # 33 .. 48 output channels: conv_split48_kernel (three 16-channel blocks, K = two taps x 16 channels)
             (3, 20, 34, 96, 48), (2, 9, 40, 256, 40), (1, 64, 48, 16, 36), (2, 96, 72, 48, 48)]


@pytest.mark.parametrize("case", CASES_3X3)
def test_split_3x3_is_as_accurate_as_the_float32_chain(ctx, lib, case):
    n, h, w, cin, cout = case
    rng = np.random.default_rng(sum(case))
    # activations spanning ~6 orders of magnitude, so that the low planes of the split matter
    x = (rng.standard_normal((n, h, w, cin)) * np.exp(2 * rng.standard_normal((n, h, w, cin)))).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    r = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    for relu, res in ((0, None), (L.PP_RELU_LAST, r), (L.PP_RELU_FIRST, r)):
        check_layer(lib, ctx, x, wt, b, 1, res=res, relu=relu)


def test_split_1x1_and_full_cover_layers(ctx, lib):
    """1x1 from 1024 input channels (ResNet's 1024 -> 256 and strided 1024 -> 2048) and the RoI head's fc6, a 7x7 'valid'
    convolution over a 7x7 input = one product over 49 * cin channels"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 12, 20, 1024)).astype(np.float32)
    wt = (rng.standard_normal((256, 1024, 1, 1)) / 32).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    r = rng.standard_normal((2, 12, 20, 256)).astype(np.float32)
    check_layer(lib, ctx, x, wt, b, 0, res=r, relu=L.PP_RELU_LAST)
    wt2 = (rng.standard_normal((96, 1024, 1, 1)) / 32).astype(np.float32)
    check_layer(lib, ctx, x, wt2, rng.standard_normal(96).astype(np.float32), 0, stride=2)
    xr = rng.standard_normal((300, 7, 7, 64)).astype(np.float32)           # 300 RoIs (ragged last tile)
    wf = (rng.standard_normal((128, 64, 7, 7)) / 56).astype(np.float32)
    check_layer(lib, ctx, xr, wf, rng.standard_normal(128).astype(np.float32), 0, relu=L.PP_RELU_LAST)


def test_split_activation_epilogues(ctx, lib):
    """LeakyReLU / Mish / ELU / Swish epilogues (YOLOv4, mars-small128, YOLOX): same double-precision evaluation as the fp32
    kernels, on the split kernel's sums"""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 26, 26, 64)).astype(np.float32)
    wt = (rng.standard_normal((128, 64, 3, 3)) / 24).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    r = rng.standard_normal((2, 26, 26, 128)).astype(np.float32)
    for act in (L.PP_ACT_LEAKY, L.PP_ACT_MISH, L.PP_ACT_ELU, L.PP_ACT_SWISH):
        for res in (None, r):
            exact, split = both(lib, lambda: hip_conv_op(ctx, x, wt, b, pad=(1, 1), relu=act, res1=res))
            assert np.isfinite(split).all() and not np.array_equal(exact, split)
            assert np.abs(split - exact).max() <= 1e-5 * np.abs(exact).max(), act


def test_split_eight_wave_forms(ctx, lib):
    """layers with >= 16 channel chunks and >= 512 workgroups take the 8-wave form (512-pixel tiles, weights through an LDS ring
    filled by the DMA path): tile form with two channel blocks per wave, with one (Cout = 96), ragged maps, and the stream form
    on zero-halo buffers -- each against the bit-exact kernel (the float64 comparison of the small cases covers the arithmetic)"""
    rng = np.random.default_rng(21)
    for n, h, w, cin, cout in ((8, 96, 96, 256, 256), (10, 96, 96, 256, 96), (9, 90, 100, 272, 128)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        r = rng.standard_normal((n, h, w, cout)).astype(np.float32)
        exact, split = both(lib, lambda: hip_conv_op(ctx, x, wt, b, pad=(1, 1), relu=L.PP_RELU_LAST, res1=r))
        scale = np.abs(exact).max()
        assert np.isfinite(split).all() and not np.array_equal(exact, split)
        assert np.abs(split - exact).max() <= 1e-5 * scale, (n, h, w, cin, cout, np.abs(split - exact).max() / scale)
    # stream form: conv -> conv on a halo buffer, 40x40 maps, 48 images
    c = 256
    pb = ProgramBuilder()
    xin_b = pb.buf(40, 40, c, name="input")
    w = [(rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32) for _ in range(2)]
    bb = [rng.standard_normal(c).astype(np.float32) for _ in range(2)]
    y1 = pb.conv(xin_b, w[0], bb[0], pad=1, relu=L.PP_RELU_LAST)
    out = pb.buf(40, 40, c, name="output")
    pb.conv(y1, w[1], bb[1], pad=1, relu=L.PP_RELU_LAST, out=out)
    prog = pb.build()
    assert max(prog.buf_pad) > 0
    xin = rng.standard_normal((48, 40, 40, c)).astype(np.float32)
    exact, split = both(lib, lambda: Net(ctx, prog, max_batch=48).forward(xin))
    assert not np.array_equal(exact, split) and np.abs(split - exact).max() <= 1e-5 * np.abs(exact).max()


@pytest.mark.parametrize("halo", ["1", "0"])
def test_split_chain_on_zero_halo_buffers(ctx, lib, monkeypatch, halo):
    """conv -> conv -> conv + residual through a layer program: with the planner's zero-halo buffers the 3x3 layers run the
    split kernel's stream form (tiles of 256 consecutive positions of the padded tensor), without them its tile form; the halo
    must still be zero afterwards (a second run gives the same result)"""
    monkeypatch.setenv("POSEPIPE_CONV_HALO", halo)
    rng = np.random.default_rng(11)
    c = 48
    pb = ProgramBuilder()
    x = pb.buf(24, 36, c, name="input")
    w = [(rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32) for _ in range(3)]
    b = [rng.standard_normal(c).astype(np.float32) for _ in range(3)]
    y1 = pb.conv(x, w[0], b[0], pad=1, relu=L.PP_RELU_LAST)
    y2 = pb.conv(y1, w[1], b[1], pad=1, relu=L.PP_RELU_LAST)
    out = pb.buf(24, 36, c, name="output")
    pb.conv(y2, w[2], b[2], pad=1, relu=L.PP_RELU_LAST, res1=y1, out=out)
    prog = pb.build()
    assert (max(prog.buf_pad) > 0) == (halo == "1")
    xin = rng.standard_normal((5, 24, 36, c)).astype(np.float32)

    def run():
        net = Net(ctx, prog, max_batch=5)
        a = net.forward(xin)
        assert np.array_equal(net.forward(xin), a)
        return a
    exact, split = both(lib, run)
    ref = xin
    a1 = np.maximum(conv64(ref, w[0], b[0], 1), 0)
    a2 = np.maximum(conv64(a1, w[1], b[1], 1), 0)
    a3 = np.maximum(conv64(a2, w[2], b[2], 1) + a1, 0)
    scale = np.abs(a3).max()
    assert not np.array_equal(exact, split)
    assert np.abs(split - a3).max() <= 1.25 * np.abs(exact - a3).max() + 1e-7 * scale
    assert np.abs(split - exact).max() <= 1e-5 * scale


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(a).max()


def test_split_backbones_agree_with_the_bit_exact_kernels(ctx, lib):
    """HRNet-W32 / W48, the detector's image program (ResNet-50 + FPN + RPN head) and its RoI head, the ReID ResNet-50, YOLOv4
    (Mish / LeakyReLU), mars-small128 (ELU), YOLOX (Swish): every named output within 2e-5 of the output range of the bit-exact
    run (measured 2..7e-6)"""
    from posepipeline_amd.models import reid_r50
    rng = np.random.default_rng(1)
    progs = {}
    for name, spec in (("w32", hrnet.HRNetSpec(32, 17, 128, 96)), ("w48", hrnet.HRNetSpec(48, 17, 128, 96))):
        progs[name] = (hrnet.build_hrnet_program(spec, synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)), "input", 3)
    dsd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    progs["det"] = (fr.build_image_program(dsd, 160, 288), "input", 2)
    progs["roi"] = (fr.build_roi_program(dsd), "roi_in", 600)
    progs["reid"] = (reid_r50.build_reid_program(synth.synth_state_dict(reid_r50.reid_param_shapes(), seed=7)), "input", 4)
    from posepipeline_amd.models import mars, yolov4, yolox
    progs["yolov4"] = (yolov4.build_yolov4_program(yolov4.synth_params(yolov4.yolov4_param_shapes(), seed=4), size=224), "input", 2)
    progs["mars"] = (mars.build_mars_program(yolov4.synth_params(mars.mars_param_shapes(), seed=5)), "input", 16)
    progs["yolox"] = (yolox.build_yolox_program(synth.synth_state_dict(yolox.yolox_param_shapes(), seed=6), 160, 256), "input", 1)
    for name, (prog, in_name, batch) in progs.items():
        x = rng.standard_normal((batch,) + tuple(prog.bufs[prog.named[in_name]])).astype(np.float32)
        outs = [k for k in prog.named if k != in_name]

        def run():
            net = Net(ctx, prog, max_batch=batch)
            return {k: net.forward(x, in_name=in_name, out_name=k) for k in outs}
        exact, split = both(lib, run)
        changed = False
        for k in outs:
            assert np.isfinite(split[k]).all()
            assert _rel(exact[k], split[k]) <= 2e-5, (name, k, _rel(exact[k], split[k]))
            changed |= not np.array_equal(exact[k], split[k])
        assert changed, name


def _hrnet_keypoints(ctx, spec, sd, x):
    """heat-maps [n][K][h][w] and DARK-decoded keypoints of a batch of crops"""
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=x.shape[0])
    hh, hw = spec.heatmap_hw
    hm = net.forward(x).reshape(x.shape[0], spec.num_joints, hh, hw)            # the head writes NCHW
    cs = np.tile(np.array([[320.0, 240.0, 1.2, 1.6]], np.float32), (hm.shape[0], 1))
    kp, _ = ops.flip_merge_decode(ctx, hm, None, cs, post="unbiased", blur_kernel=11)
    return hm, kp


def test_split_keypoints_move_no_more_than_under_a_float32_reordering(ctx, lib):
    """What 1e-3 px can and cannot mean with seeded random weights: the heat-maps are noise, so the argmax of a joint can sit
    between two near-equal maxima and DARK's Taylor step divides by a near-singular Hessian -- float32 ROUNDING NOISE of the
    network (1e-6 of the heat-map range) then moves some joints by far more than 1e-3 px, whichever kernel computes it.
    Control: the bit-exact kernels on the TRANSPOSED problem (inputs and every kernel transposed, output transposed back):
    mathematically the same network, but the float32 FMA chain visits the taps in another order -- an equally valid float32
    evaluation, e.g. what separates the oracle from the reference's own BLAS.  The split kernels must not move heat-maps or
    joints more than that control does."""
    spec = hrnet.HRNetSpec(32, 17, 128, 96)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    rng = np.random.default_rng(12)
    x = np.zeros((48, 128, 96, 4), np.float32)
    x[..., :3] = rng.standard_normal((48, 128, 96, 3))
    L.check(lib.pp_conv_exact(1), "pp_conv_exact")
    hm_e, kp_e = _hrnet_keypoints(ctx, spec, sd, x)
    tspec = hrnet.HRNetSpec(32, 17, 96, 128)
    tsd = {k: (np.ascontiguousarray(np.transpose(v, (0, 1, 3, 2))) if np.ndim(v) == 4 else v) for k, v in sd.items()}
    tnet = Net(ctx, hrnet.build_hrnet_program(tspec, tsd), max_batch=48)
    hm_c = tnet.forward(np.ascontiguousarray(np.transpose(x, (0, 2, 1, 3)))).reshape(48, 17, 24, 32)
    hm_c = np.ascontiguousarray(np.transpose(hm_c, (0, 1, 3, 2)))
    cs = np.tile(np.array([[320.0, 240.0, 1.2, 1.6]], np.float32), (48, 1))
    kp_c, _ = ops.flip_merge_decode(ctx, hm_c, None, cs, post="unbiased", blur_kernel=11)
    L.check(lib.pp_conv_exact(0), "pp_conv_exact")
    hm_s, kp_s = _hrnet_keypoints(ctx, spec, sd, x)
    L.check(lib.pp_conv_exact(1), "pp_conv_exact")
    rng_hm = np.abs(hm_e).max()
    d_ctrl_hm, d_split_hm = np.abs(hm_c - hm_e).max() / rng_hm, np.abs(hm_s - hm_e).max() / rng_hm
    assert 0 < d_ctrl_hm <= 2e-5 and 0 < d_split_hm <= 2e-5
    r_ctrl_hm, r_split_hm = np.sqrt(np.mean((hm_c - hm_e) ** 2)) / rng_hm, np.sqrt(np.mean((hm_s - hm_e) ** 2)) / rng_hm
    print(f"heat-maps vs bit-exact: reordered max {d_ctrl_hm:.2e} rms {r_ctrl_hm:.2e} | split max {d_split_hm:.2e} rms {r_split_hm:.2e}")
    # heat-maps: the noise level of a reordering (which here only permutes the 9 taps, not the channel sums: a lower bound)
    assert d_split_hm <= 2.5 * d_ctrl_hm and r_split_hm <= 2.5 * r_ctrl_hm, (d_split_hm, d_ctrl_hm, r_split_hm, r_ctrl_hm)
    dc = np.abs(kp_c[:, :, :2] - kp_e[:, :, :2]).max(axis=2).ravel()
    ds = np.abs(kp_s[:, :, :2] - kp_e[:, :, :2]).max(axis=2).ravel()
    bad_s, bad_c = int((ds > TOL_PX).sum()), int((dc > TOL_PX).sum())
    print(f"joints ({ds.size}): reordered median {np.median(dc):.1e} p90 {np.percentile(dc, 90):.1e} >1e-3px {bad_c} max {dc.max():.2g} | "
          f"split median {np.median(ds):.1e} p90 {np.percentile(ds, 90):.1e} >1e-3px {bad_s} max {ds.max():.2g}")
    assert np.median(ds) <= max(1e-4, 2.5 * np.median(dc)), (np.median(ds), np.median(dc))
    assert np.percentile(ds, 90) <= max(TOL_PX, 2.5 * np.percentile(dc, 90)), (np.percentile(ds, 90), np.percentile(dc, 90))
    assert bad_s <= 2.5 * bad_c + 5, (bad_s, bad_c)                  # joints beyond 1e-3 px: the control has them, too


def test_split_keypoints_within_1e_3_px_on_peaked_heatmaps(ctx, lib):
    """The same question on WELL-CONDITIONED heat-maps, which is what a trained network produces: a small program of 3x3 layers with
    positive (smoothing) weights turns blob images into smooth single-peaked maps; then every joint decoded from the split
    kernels' maps is within north_star's 1e-3 px of the bit-exact kernels' (measured ~1e-5), arg-max and DARK refinement included."""
    rng = np.random.default_rng(41)
    h, w, c, k, n = 64, 48, 32, 17, 24
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    x = np.zeros((n, h, w, c), np.float32)
    for i in range(n):
        for ch in range(c):
            cy, cx, sg = rng.uniform(12, h - 12), rng.uniform(10, w - 10), rng.uniform(2.0, 4.0)
            x[i, :, :, ch] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg)) * rng.uniform(0.5, 1.5)
    x += rng.uniform(0, 1e-3, x.shape).astype(np.float32)
    pb = ProgramBuilder()
    t = pb.buf(h, w, c, name="input")
    for _ in range(3):
        wt = np.abs(rng.standard_normal((c, c, 3, 3))).astype(np.float32)
        wt /= wt.sum(axis=(1, 2, 3), keepdims=True)
        t = pb.conv(t, wt, np.zeros(c, np.float32), pad=1, relu=L.PP_RELU_LAST)
    # head: each joint follows a few feature channels (so its map has one dominant blob)
    wh = np.zeros((k, c, 1, 1), np.float32)
    for j in range(k):
        wh[j, rng.choice(c, 2, replace=False), 0, 0] = (1.0, 0.15)
    out = pb.buf(h, w, k, name="output")
    pb.conv(t, wh, np.zeros(k, np.float32), pad=0, out=out, out_nchw=True)
    prog = pb.build()
    cs = np.tile(np.array([[320.0, 240.0, 1.2, 1.6]], np.float32), (n, 1))

    def run():
        hm = Net(ctx, prog, max_batch=n).forward(x).reshape(n, k, h, w)
        kp, _ = ops.flip_merge_decode(ctx, hm, None, cs, post="unbiased", blur_kernel=11)
        return hm, kp
    (hm_e, kp_e), (hm_s, kp_s) = both(lib, run)
    assert not np.array_equal(hm_e, hm_s) and np.abs(hm_s - hm_e).max() <= 1e-5 * np.abs(hm_e).max()
    top2 = np.sort(hm_e.reshape(n, k, -1), axis=2)[:, :, -2:]
    assert (top2[:, :, 1] > 0).all()
    d = np.abs(kp_s[:, :, :2] - kp_e[:, :, :2]).max(axis=2)
    print(f"peaked heat-maps: joints {d.size}, max deviation {d.max():.2e} px, scores {np.abs(kp_s[:, :, 2] - kp_e[:, :, 2]).max():.2e}")
    assert d.max() <= TOL_PX, d.max()
    assert np.abs(kp_s[:, :, 2] - kp_e[:, :, 2]).max() <= 1e-5


def test_split_topdown_stage_close_to_exact(ctx, lib):
    """BASELINE.json configs[1] shape through the fused stage: HRNet-W32 256x192, 64 person crops + flip test -> DARK decode.
    Scores within 1e-5; joints: at least 90 % within 1e-3 px of the bit-exact path (the rest: the conditioning shown above)"""
    spec = hrnet.hrnet_w32_256x192()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    frames, bboxes = synth_clip(np.random.default_rng(4), 64, 480, 640)
    idx = np.arange(64, dtype=np.int32)

    def run():
        net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=128)
        td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="unbiased", blur_kernel=11)
        kp, valid = td.run(frames, idx, bboxes)
        return kp
    exact, split = both(lib, run)
    assert exact.shape == (64, 17, 3) and not np.array_equal(exact, split)
    d = np.abs(exact[:, :, :2] - split[:, :, :2]).max(axis=2).ravel()
    assert (d <= TOL_PX).mean() >= 0.9, (d <= TOL_PX).mean()
    assert np.median(d) <= 1e-4
    assert np.abs(exact[:, :, 2] - split[:, :, 2]).max() <= 1e-5 * max(np.abs(exact[:, :, 2]).max(), 1e-6) + 1e-7


def test_split_detector_same_boxes(ctx, lib):
    """Faster-RCNN end to end (image program, RPN top-k + NMS, RoIAlign, RoI head, per-class NMS): the same detections in the
    detections up to near-ties of top-k / NMS (>= 90 % of them re-found), scores within 1e-4.  Boxes: with seeded random
    weights the regression deltas are O(1) on boxes hundreds of pixels wide and go through exp(), so float32 rounding noise
    of fc6 (12544-term sums) shows as ~1e-2 px: bound 0.05 px (5e-5 of the box size), not north_star's 1e-3, which presumes
    trained weights"""
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    rng = np.random.default_rng(2)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(3)])

    def run():
        det = fr.Detector(ctx, sd, 135, 240, max_frames=3)
        return det.run(frames)
    exact, split = both(lib, run)
    total = sum(len(e) for e in exact)
    assert total > 0
    close = 0
    for e, s in zip(exact, split):
        assert abs(len(e) - len(s)) <= max(1, len(e) // 10)           # a near-tie in top-k / NMS may swap a candidate
        for row in e:
            if len(s) and (np.abs(s[:, :4] - row[:4]).max(axis=1) <= 0.05).any():
                j = int(np.argmin(np.abs(s[:, :4] - row[:4]).max(axis=1)))
                close += abs(s[j, 4] - row[4]) <= 1e-4
    assert close >= 0.9 * total, (close, total)


def test_split_cascade_same_tracks_close_joints(ctx, lib):
    """detect -> track -> 2D -> 3D over a chunked clip: identical track ids and frame indices; 2D joints: median within 1e-4 px,
    90 % within 1e-3 px of the bit-exact cascade (random-weight conditioning, see above); 3D within 0.02 mm (VideoPose3D's
    unit is the metre)"""
    from posepipeline_amd.cascade import Cascade
    rng = np.random.default_rng(3)
    h, w, n = 135, 240, 6
    frames = np.stack([synth_frame(rng, h, w) for _ in range(n)])
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_spec = hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(n)]

    def run():
        cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=3, max_persons=1, pose_spec=pose_spec)
        outs = [cas.step(frames[0:3], replay=gt[0:3]), cas.step(frames[3:6], replay=gt[3:6]), cas.flush()]
        return outs
    e_out, s_out = both(lib, run)
    assert [[[r[0] for r in f] for f in o["tracks"]] for o in e_out] == [[[r[0] for r in f] for f in o["tracks"]] for o in s_out]
    d2, d3 = [], []
    for eo, so in zip(e_out, s_out):
        assert eo["keypoints"].keys() == so["keypoints"].keys() and eo["keypoints_3d"].keys() == so["keypoints_3d"].keys()
        for tid in eo["keypoints"]:
            assert eo["keypoints_frames"][tid].tolist() == so["keypoints_frames"][tid].tolist()
            d2.append(np.abs(eo["keypoints"][tid][:, :, :2] - so["keypoints"][tid][:, :, :2]).max(axis=2).ravel())
        for tid in eo["keypoints_3d"]:
            assert eo["keypoints_3d_frames"][tid].tolist() == so["keypoints_3d_frames"][tid].tolist()
            d3.append(np.abs(eo["keypoints_3d"][tid] - so["keypoints_3d"][tid]).ravel())
    d2, d3 = np.concatenate(d2), np.concatenate(d3)
    assert d2.size == n * 17 and d3.size == n * 17 * 3
    assert np.median(d2) <= 1e-4 and (d2 <= TOL_PX).mean() >= 0.9 and d2.max() <= 0.05, (np.median(d2), d2.max())
    # the lifting network only sees the 2D differences above
    assert d3.max() <= 2e-5, (d3.max(), d2.max())           # measured 6e-6 m for 1.6e-3 px of 2D difference (random lifting weights)


def test_split_roi_align_separable_form(ctx, lib):
    """Default numerics: RoIAlign is evaluated separably (row sums shared by the samples that touch a row, roi_align_sep_kernel) --
    the same samples, validity rule and clamping as the sample loop of mmcv / oracle.detector.roi_align, another float32
    summation order.  On the device's OWN FPN features and proposals: every output within 1e-5 of the feature range of the
    oracle's loop, for RoIs of every FPN level, RoIs that leave the image on every side, tiny and very wide ones."""
    from oracle import detector as odet
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    for k, g in (("detector.rpn_head.rpn_cls.weight", 0.5), ("detector.rpn_head.rpn_reg.weight", 0.1)):
        sd[k] = (sd[k] * g).astype(np.float32)
    rng = np.random.default_rng(2)
    frame = synth_frame(rng, 135, 240)
    L.check(lib.pp_conv_exact(0), "pp_conv_exact")
    det = fr.Detector(ctx, sd, 135, 240, max_frames=1)
    L.check(lib.pp_conv_exact(1), "pp_conv_exact")
    assert det.net_b.numerics == "split"
    _, props = det.run(frame[None], want_proposals=True)
    props = props[0]
    feats = [det.net_a.read(f"p{i}", 1) for i in range(2, 6)]
    got = det.net_b.read("roi_in", len(props))
    lv = odet.map_roi_levels(props)
    w, h = props[:, 2] - props[:, 0], props[:, 3] - props[:, 1]
    pick = set(range(12))
    for level in range(4):
        pick |= set(np.flatnonzero(lv == level)[:12].tolist())
    pick |= set(np.argsort(w)[:6].tolist()) | set(np.argsort(-w)[:6].tolist()) | set(np.argsort(-w / np.maximum(h, 1e-3))[:6].tolist())
    pick |= set(np.argsort(props[:, 0])[:4].tolist()) | set(np.argsort(props[:, 1])[:4].tolist())
    pick |= set(np.argsort(-props[:, 2])[:4].tolist()) | set(np.argsort(-props[:, 3])[:4].tolist())
    pick = sorted(pick)
    assert len({int(l) for l in lv[pick]}) >= 3
    scale = max(float(np.abs(f_).max()) for f_ in feats)
    worst = 0.0
    for i in pick:
        ref = odet.roi_align(feats[lv[i]][0], props[i], 1.0 / odet.STRIDES[lv[i]])
        worst = max(worst, float(np.abs(got[i] - ref).max()))
        assert np.abs(got[i] - ref).max() <= 1e-5 * scale, (i, props[i], lv[i], np.abs(got[i] - ref).max() / scale)
    print(f"separable RoIAlign: {len(pick)} RoIs, max deviation {worst / scale:.2e} of the feature range")
    # RoIs past n_rois are zero rows
    assert not det.net_b.read("roi_in", det.MAX_ROIS)[len(props):].any() or len(props) == det.MAX_ROIS


# n, h, w, cin, cout: 3x3 stride 2 (HRNet transition / fuse layers, ResNet's strided blocks) -- the product form with one step per
# (channel chunk, tap): odd and even maps (last row / column taps leave the image on the far side, too), 1..32 chunks,
# Cout with one and two channel blocks per workgroup, ragged last tile
CASES_S2 = [(2, 24, 18, 48, 96), (1, 23, 35, 16, 48), (3, 12, 10, 192, 384), (1, 40, 68, 512, 512), (2, 17, 17, 96, 32), (1, 96, 72, 48, 48),
            (2, 21, 33, 128, 128), (1, 16, 16, 256, 64)]


@pytest.mark.parametrize("case", CASES_S2)
def test_split_3x3_stride_2(ctx, lib, case, monkeypatch):
    monkeypatch.setenv("POSEPIPE_SPLIT_S2_MIN_CIN", "16")       # the form's correctness also where the launcher would not pick it
    n, h, w, cin, cout = case
    rng = np.random.default_rng(sum(case) + 7)
    x = (rng.standard_normal((n, h, w, cin)) * np.exp(2 * rng.standard_normal((n, h, w, cin)))).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    r = rng.standard_normal((n, ho, wo, cout)).astype(np.float32)
    for relu, res in ((0, None), (L.PP_RELU_LAST, r)):
        check_layer(lib, ctx, x, wt, b, 1, stride=2, res=res, relu=relu)


def test_split_stride_2_inside_a_program(ctx, lib, monkeypatch):
    """a strided 3x3 layer that READS a zero-halo buffer (its producer is a 3x3 stride-1 layer) and one with two residuals, as
    HRNet's fuse layers chain them, on ODD maps (25 x 37: the far-side taps of the last output row / column leave the image and
    must hit the halo / the validity mask): same result as the float32 MFMA kernels to 1e-5, and the split kernel -- the
    tap-gather form for the strided layers -- is what ran for EVERY layer.  POSEPIPE_SPLIT_S2_MIN_CIN is read when the net is
    created (selection), never at launch: the variable is removed again before the first launch."""
    rng = np.random.default_rng(23)
    c = 48
    pb = ProgramBuilder()
    x = pb.buf(25, 37, c, name="input")
    w = [(rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32) for co, ci in ((c, c), (96, c), (96, 96))]
    b = [rng.standard_normal(co).astype(np.float32) for co in (c, 96, 96)]
    y1 = pb.conv(x, w[0], b[0], pad=1, relu=L.PP_RELU_LAST)
    y2 = pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST)
    y3 = pb.conv(y2, w[2], b[2], pad=1, relu=L.PP_RELU_LAST)
    out = pb.buf(13, 19, 96, name="output")
    pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST, res1=y2, res2=y3, out=out)
    prog = pb.build()
    xin = rng.standard_normal((5, 25, 37, c)).astype(np.float32)

    def run():
        monkeypatch.setenv("POSEPIPE_SPLIT_S2_MIN_CIN", "16")
        net = Net(ctx, prog, max_batch=5)
        monkeypatch.delenv("POSEPIPE_SPLIT_S2_MIN_CIN")              # committed at creation: the launches must not consult it
        kinds = net.conv_kinds()
        return net.forward(xin), kinds
    (exact, k_e), (split, k_s) = both(lib, run)
    assert (k_e == 1).all() and (k_s == 2).all(), (k_e, k_s)      # strided layers included (x_pad > 0 input; res1 + res2)
    assert not np.array_equal(exact, split) and np.abs(split - exact).max() <= 1e-5 * np.abs(exact).max()


def test_f16_form_strided_patch_inside_a_program(ctx, lib16):
    """the strided-patch form where HRNet uses it: reading a ZERO-HALO buffer (producer: a 3x3 stride-1 layer), writing one, with two
    residuals, on ODD maps (49 x 73 -> 25 x 37: the last output row / column's far taps leave the image) -- large enough for the
    selection rule to pick the form (>= 400 output pixels with 96 output channels), every layer on the split kernels, same result as the
    float32 MFMA kernels to 1e-5, per-sample maxima tracked across the strided layers (samples at different magnitudes)"""
    rng = np.random.default_rng(29)
    c = 48
    pb = ProgramBuilder()
    x = pb.buf(49, 73, c, name="input")
    w = [(rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32) for co, ci in ((c, c), (96, c), (96, 96), (192, 96))]
    b = [rng.standard_normal(co).astype(np.float32) for co in (c, 96, 96, 192)]
    y1 = pb.conv(x, w[0], b[0], pad=1, relu=L.PP_RELU_LAST)
    y2 = pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST)                  # strided patch, halo in, halo out
    y3 = pb.conv(y2, w[2], b[2], pad=1, relu=L.PP_RELU_LAST)
    y4 = pb.conv(y1, w[1], b[1], pad=1, stride=2, relu=L.PP_RELU_LAST, res1=y2, res2=y3)   # two residuals
    pb.conv(y4, w[3], b[3], pad=1, stride=2, relu=L.PP_RELU_LAST, out=pb.buf(13, 19, 192, name="output"))   # 247 outputs: the tap-gather form
    prog = pb.build()
    xin = (rng.standard_normal((4, 49, 73, c)) * (10.0 ** np.arange(-2, 2)).reshape(-1, 1, 1, 1)).astype(np.float32)

    def run():
        net = Net(ctx, prog, max_batch=4)
        return net.forward(xin), net.conv_kinds()
    (exact, k_e), (split, k_s) = both(lib16, run)
    assert (k_e == 1).all() and (k_s == 2).all(), (k_e, k_s)
    scale = np.abs(exact).reshape(4, -1).max(1).reshape(4, 1, 1, 1)
    assert not np.array_equal(exact, split) and (np.abs(split - exact) <= 1e-5 * scale).all()


def test_split_1x1_with_shifted_residual(ctx, lib):
    """FPN's lateral convs: 1x1 + (top-down map read at (y >> 1, x >> 1)) -- on the product kernel the coarse residual is read with
    the shift in the epilogue, odd fine maps included"""
    rng = np.random.default_rng(61)
    for cin, h, w in ((256, 40, 68), (512, 21, 35), (1024, 10, 17)):
        x = rng.standard_normal((3, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((256, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        b = rng.standard_normal(256).astype(np.float32)
        r = rng.standard_normal((3, (h + 1) // 2, (w + 1) // 2, 256)).astype(np.float32)
        exact, split = both(lib, lambda: hip_conv_op(ctx, x, wt, b, res1=r, res1_shift=1))
        from tests.helpers import ref_conv_op
        assert np.array_equal(exact, ref_conv_op(x, wt, b, res1=r, res1_shift=1))
        assert not np.array_equal(exact, split) and np.abs(split - exact).max() <= 1e-5 * np.abs(exact).max(), cin


def test_split_program_replays_from_a_hip_graph(ctx, lib):
    """a program created with the split numerics captures into a hipGraph (no allocation, no synchronisation inside its launches:
    the split weights were built at creation) and the replay reproduces the eager run bit for bit -- HRNet-W48 layers included
    (tile / stream / ring forms, the 48-channel kernel, product kernels, one-pass fuse sums)"""
    spec = hrnet.HRNetSpec(48, 17, 128, 96)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    rng = np.random.default_rng(9)
    x = np.zeros((6, 128, 96, 4), np.float32)
    x[..., :3] = rng.standard_normal((6, 128, 96, 3))
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=6, numerics="split")
    assert (net.conv_kinds() == 2).sum() > 100
    eager = net.forward(x)
    net.capture(6)
    assert np.array_equal(net.forward(x), eager)
    assert np.array_equal(net.forward(x), eager)
    net.close()


def test_product_kernel_epilogues_agree_bit_for_bit(ctx, lib, monkeypatch):
    """conv_split_gemm_kernel stores either from the registers (a lane: 4 channels of one pixel) or transposed through LDS (whole
    128-byte lines, offsets from a per-pixel table): same arithmetic per element in the same order, so the same bits -- ragged last tiles, strides, both residuals (one read with a shift),
    ReLU first / last, halo output buffers inside a program, and fc6's full-cover product"""
    rng = np.random.default_rng(71)

    def run(fn):
        out = []
        for epi in (0, 1):
            L.check(lib.pp_debug_knob(b"split_gemm_epilogue", epi), "pp_debug_knob")
            L.check(lib.pp_conv_exact(0), "pp_conv_exact")
            out.append(fn())
        L.check(lib.pp_conv_exact(1), "pp_conv_exact")
        L.check(lib.pp_debug_knob(b"split_gemm_epilogue", -1), "pp_debug_knob")
        assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
        assert np.array_equal(out[0], out[1])

    for cin, cout, h, w, n, stride in ((1024, 256, 12, 20, 3, 1), (512, 384, 21, 35, 2, 1), (256, 1024, 17, 9, 5, 1), (1024, 512, 20, 12, 2, 2)):
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        r = rng.standard_normal((n, ho, wo, cout)).astype(np.float32)
        for relu in (0, L.PP_RELU_FIRST, L.PP_RELU_LAST):
            run(lambda: hip_conv_op(ctx, x, wt, b, stride=stride, relu=relu, res1=r))
    x = rng.standard_normal((3, 21, 35, 512)).astype(np.float32)                      # shifted residual (FPN lateral)
    wt = (rng.standard_normal((256, 512, 1, 1)) / np.sqrt(512)).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    r = rng.standard_normal((3, 11, 18, 256)).astype(np.float32)
    run(lambda: hip_conv_op(ctx, x, wt, b, res1=r, res1_shift=1))
    xr = rng.standard_normal((300, 7, 7, 64)).astype(np.float32)                       # fc6 form
    wf = (rng.standard_normal((128, 64, 7, 7)) / 56).astype(np.float32)
    bf = rng.standard_normal(128).astype(np.float32)
    run(lambda: hip_conv_op(ctx, xr, wf, bf, relu=L.PP_RELU_LAST))
    # inside a program: the 1x1 writes a zero-halo buffer that a 3x3 reads (y_pad > 0), second residual
    pb = ProgramBuilder()
    xin_b = pb.buf(20, 12, 512, name="input")
    y1 = pb.conv(xin_b, (rng.standard_normal((256, 512, 1, 1)) / 23).astype(np.float32), rng.standard_normal(256).astype(np.float32), relu=L.PP_RELU_LAST)
    out = pb.buf(20, 12, 256, name="output")
    pb.conv(y1, (rng.standard_normal((256, 256, 3, 3)) / 48).astype(np.float32), rng.standard_normal(256).astype(np.float32), pad=1, res1=y1, out=out)
    prog = pb.build()
    xin = rng.standard_normal((7, 20, 12, 512)).astype(np.float32)

    def net_run():
        net = Net(ctx, prog, max_batch=7, numerics="split")
        y = net.forward(xin)
        net.close()
        return y
    run(net_run)


# ---- fp16 form (round 5): the activation scale follows the data, per sample ------------------------------------------------------
@pytest.fixture
def lib16(ctx):
    L.check(ctx.lib.pp_conv_split_kind(1), "pp_conv_split_kind")
    yield ctx.lib
    L.check(ctx.lib.pp_conv_split_kind(-1), "pp_conv_split_kind")
    L.check(ctx.lib.pp_conv_exact(1), "pp_conv_exact")


# (n, h, w, cin, cout, k, stride): tap kernel (tile / stream), 48-channel form, 8-wave form, one-tap product, product kernel, tap-gather
CASES_F16 = [(2, 24, 18, 64, 64, 3, 1), (3, 24, 18, 48, 48, 3, 1), (1, 40, 68, 256, 256, 3, 1), (8, 96, 96, 256, 256, 3, 1),
             (2, 12, 20, 1024, 96, 1, 2), (2, 40, 68, 256, 1024, 1, 1), (2, 40, 68, 128, 128, 3, 2)]
MAGNITUDES = [1e-30, 1e-6, 1.0, 3e3, 1e7]


@pytest.mark.parametrize("case", CASES_F16)
def test_f16_form_keeps_the_yardstick_at_any_magnitude(ctx, lib16, case):
    """the fp32 yardstick (error against a float64 convolution <= the float32 kernel's, x 1.25 on the rms) with
    inputs from 1e-30 to 1e7 -- far below float16's smallest normal and far above its largest value -- and with every SAMPLE of a
    batch at its own magnitude (1e-6 .. 1e4): the scale is taken per sample from the tensor's running maximum, nothing saturates
    and nothing underflows.  Weights: per-channel ranges over ~8 orders of magnitude (BatchNorm-folded kernels)."""
    n, h, w, cin, cout, k, stride = case
    rng = np.random.default_rng(sum(case) + 3)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    wt *= np.exp(3 * rng.standard_normal((cout, 1, 1, 1))).astype(np.float32)
    pad = k // 2
    mags = [np.full(n, m) for m in MAGNITUDES] + [10.0 ** np.linspace(-6, 4, n)]
    for mag in mags:
        x = (rng.standard_normal((n, h, w, cin)) * np.exp(rng.standard_normal((n, h, w, cin))) * mag.reshape(-1, 1, 1, 1)).astype(np.float32)
        b = (rng.standard_normal(cout) * np.abs(x).mean()).astype(np.float32)
        ref = conv64(x, wt, b, pad, stride)
        exact, split = both(lib16, lambda: hip_conv_op(ctx, x, wt, b, stride=stride, pad=(pad, pad)))
        assert np.isfinite(split).all() and not np.array_equal(exact, split)
        # per sample and per output channel (their ranges differ by orders of magnitude)
        scale = np.abs(ref).reshape(n, -1, cout).max(1).reshape(n, 1, 1, cout) + 1e-300
        rms = lambda y: float(np.sqrt(np.mean(((y - ref) / scale) ** 2)))
        assert rms(split) <= 1.25 * rms(exact) + 1e-9, (mag[0], rms(split), rms(exact))
        assert np.abs((split - ref) / scale).max() <= 1.5 * np.abs((exact - ref) / scale).max() + 1e-7


def test_f16_form_channel_ranges_spread_over_2_pow_20(ctx, lib16):
    """the activation scale is ONE power of two per sample of the whole tensor.  With input channels whose ranges are
    spread over 2^20 (BN-folded nets can produce that) the small channels sit in float16's subnormals for the residual term: what
    the form guarantees there is ABSOLUTE -- every input element is represented to <= 2^-22 of itself or 2^-39 of the sample's
    maximum m, whichever is larger -- so the output error is bounded by sum_k |w_k| * max(2^-22 |x_k|, 2^-39 m) plus the float32
    accumulation's own rounding (include/posepipe_hip.h, pp_conv_split_kind).  Asserted against a float64 convolution; the
    relative error of an output fed ONLY by tiny channels may exceed float32's, which is the documented limit."""
    n, h, w, cin, cout = 2, 24, 18, 64, 64
    rng = np.random.default_rng(77)
    ch = (2.0 ** np.linspace(0, -20, cin)).astype(np.float32)                  # channel c lives at 2^(-20 c / 63) of channel 0
    x = (rng.standard_normal((n, h, w, cin)).astype(np.float32) * ch).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    ref = conv64(x, wt, b, 1)
    exact, split = both(lib16, lambda: hip_conv_op(ctx, x, wt, b, stride=1, pad=(1, 1)))
    m = np.abs(x).reshape(n, -1).max(1).reshape(n, 1, 1, 1).astype(np.float64)
    rep = np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -39 * m)   # representation bound per input element
    wrep = 2.0 ** -22 * np.abs(wt.astype(np.float64))                             # weights: 22 bits per output channel's own range
    bound = conv64(rep, np.abs(wt), b, 1) + conv64(np.abs(x), wrep, b, 1) + 2.0 ** -23 * 24 * conv64(np.abs(x), np.abs(wt), b, 1)
    assert np.isfinite(split).all()
    assert (np.abs(split - ref) <= bound + 1e-30).all(), float((np.abs(split - ref) / (bound + 1e-30)).max())
    # outputs dominated by the big channels: the usual yardstick holds
    scale = np.abs(ref).max()
    assert np.sqrt(np.mean((split - ref) ** 2)) <= 1.25 * np.sqrt(np.mean((exact - ref) ** 2)) + 1e-9 * scale
    # a layer that reads ONLY the small half of the channels (weights of the big ones zero): absolute bound still holds, and the
    # error relative to THAT output's range is what the header warns about -- measured and printed, not asserted small
    wt2 = wt.copy()
    wt2[:, :48] = 0
    ref2 = conv64(x, wt2, b, 1)
    _, split2 = both(lib16, lambda: hip_conv_op(ctx, x, wt2, b, stride=1, pad=(1, 1)))
    bound2 = conv64(rep, np.abs(wt2), b, 1) + conv64(np.abs(x), 2.0 ** -22 * np.abs(wt2.astype(np.float64)), b, 1) + 2.0 ** -23 * 24 * conv64(np.abs(x), np.abs(wt2), b, 1)
    assert (np.abs(split2 - ref2) <= bound2 + 1e-30).all()
    print(f"[f16 form, channels below 2^-15 of the sample maximum only] error {np.abs(split2 - ref2).max() / np.abs(ref2).max():.2e} of that output's range "
          f"(float32 chain: ~1e-7); bound held with max ratio {float((np.abs(split2 - ref2) / (bound2 + 1e-30)).max()):.2f}")


def _chain_program(rng, c=64, h=20, w=12):
    """3x3 -> 1x1 (product kernel, 128 output channels) -> 3x3 stride 2 -> 3x3 with a residual: every tracked-maximum path of a
    program (fused epilogues of the tap / product kernels, zero-halo buffers, an external input)"""
    pb = ProgramBuilder()
    x = pb.buf(h, w, c, name="input")
    mk = lambda co, ci, k: ((rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32), (0.1 * rng.standard_normal(co)).astype(np.float32))
    w1, b1 = mk(128, c, 3)
    w2, b2 = mk(128, 128, 1)
    w3, b3 = mk(128, 128, 3)
    w4, b4 = mk(128, 128, 3)
    y1 = pb.conv(x, w1, b1, pad=1, relu=L.PP_RELU_LAST)
    y2 = pb.conv(y1, w2, b2, pad=0, relu=L.PP_RELU_LAST, res1=y1)
    y3 = pb.conv(y2, w3, b3, pad=1, stride=2, relu=L.PP_RELU_LAST)
    pb.conv(y3, w4, b4, pad=1, relu=L.PP_RELU_LAST, res1=y3, out=pb.buf(h // 2, w // 2, 128, name="output"))
    return pb.build()


def test_f16_form_results_do_not_depend_on_the_batch(ctx, lib16, monkeypatch):
    """A sample's result is the same BITS whatever else is in the batch (the scale is per sample, the running maxima are per
    sample): six samples whose magnitudes spread over 1e-8 .. 1e8 run together, alone, and in reversed order.  (What
    tests/test_gpu_sharded.py relies on when it compares shards with the whole clip.)"""
    monkeypatch.setenv("POSEPIPE_SPLIT_S2_MIN_CIN", "16")
    rng = np.random.default_rng(77)
    prog = _chain_program(rng)
    x = (rng.standard_normal((6, 20, 12, 64)) * (10.0 ** np.linspace(-8, 8, 6)).reshape(-1, 1, 1, 1)).astype(np.float32)
    net = Net(ctx, prog, max_batch=6, numerics="split_f16")
    assert (net.conv_kinds() == 2).all() and net.split_kind == "split_f16"
    together = net.forward(x)
    assert np.isfinite(together).all()
    for i in range(6):
        assert np.array_equal(net.forward(x[i:i + 1])[0], together[i]), i
    assert np.array_equal(net.forward(x[::-1].copy())[::-1], together)
    # ... and close to the float32 kernels at every magnitude
    exact = Net(ctx, prog, max_batch=6, numerics="exact").forward(x)
    for i in range(6):
        assert np.abs(together[i] - exact[i]).max() <= 1e-5 * np.abs(exact[i]).max(), i


def test_f16_form_non_finite_input_stays_in_its_sample(ctx, lib16, monkeypatch):
    """inf / NaN in one sample: that sample's outputs are unspecified (an infinite maximum collapses its scale, and fmaxf-style ReLUs
    swallow NaNs on either kernel family); every OTHER sample of the batch is untouched, bit for bit"""
    monkeypatch.setenv("POSEPIPE_SPLIT_S2_MIN_CIN", "16")
    rng = np.random.default_rng(78)
    prog = _chain_program(rng)
    x = rng.standard_normal((4, 20, 12, 64)).astype(np.float32)
    net = Net(ctx, prog, max_batch=4, numerics="split_f16")
    clean = net.forward(x)
    bad = x.copy()
    bad[1, 3, 4, 5] = np.inf
    bad[2, 7, 2, 9] = np.nan
    out = net.forward(bad)
    for i in (0, 3):
        assert np.array_equal(out[i], clean[i])
    # the maxima are reset per run: the next clean run is clean again
    assert np.array_equal(net.forward(x), clean)


# (n, h, w, cin, cout): 3x3 stride 2 on the STRIDED-PATCH form (round 6, fp16 form; output maps >= 1500 pixels, or >= 400 with >= 96 output
# channels): odd and even maps on both axes (the last output row / column's far taps leave the image), 1 .. 4 channel blocks per wave
# (32 / 64 / 96 / 128 output channels, and 48 = one and a half), 1 .. 16 channel chunks, every tile shape the picker has (4x32, 8x16, 16x8)
CASES_S2P = [(2, 96, 72, 48, 96), (1, 96, 72, 48, 48), (2, 95, 71, 64, 64), (1, 81, 135, 128, 128), (1, 80, 136, 256, 256), (3, 47, 37, 96, 192),
             (1, 192, 144, 64, 64), (2, 60, 100, 16, 32), (1, 40, 68, 256, 96), (1, 131, 33, 32, 160)]


@pytest.mark.parametrize("case", CASES_S2P)
def test_f16_form_3x3_stride_2_strided_patch(ctx, lib16, case):
    """conv_split_kernel<.., S2>: the patch of a 128-pixel output tile de-interleaved into its four phases in LDS -- the yardstick of
    every other form (error against a float64 convolution <= the float32 kernel's) with and without ReLU / residual, and the
    per-sample scale (samples at different magnitudes)"""
    n, h, w, cin, cout = case
    rng = np.random.default_rng(sum(case) + 11)
    x = (rng.standard_normal((n, h, w, cin)) * np.exp(2 * rng.standard_normal((n, h, w, cin))) *
         (10.0 ** np.linspace(-3, 2, n)).reshape(-1, 1, 1, 1)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = (rng.standard_normal(cout) * np.abs(x).mean()).astype(np.float32)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    r = (rng.standard_normal((n, ho, wo, cout)) * np.abs(x).mean()).astype(np.float32)
    for relu, res in ((0, None), (L.PP_RELU_LAST, r)):
        ref = conv64(x, wt, b, 1, 2, res, relu)
        exact, split = both(lib16, lambda: hip_conv_op(ctx, x, wt, b, stride=2, pad=(1, 1), relu=relu, res1=res))
        assert np.isfinite(split).all() and split.shape == ref.shape and not np.array_equal(exact, split)
        scale = np.abs(ref).reshape(n, -1).max(1).reshape(n, 1, 1, 1) + 1e-300
        rms = lambda y: float(np.sqrt(np.mean(((y - ref) / scale) ** 2)))
        assert rms(split) <= 1.25 * rms(exact) + 1e-9, (rms(split), rms(exact))
        assert np.abs((split - ref) / scale).max() <= 1.5 * np.abs((exact - ref) / scale).max() + 1e-7


# (n, h, w): whole tiles, ragged tiles on both axes, a map smaller than one tile, the smoke-size frame, many tiles per workgroup
CASES_STEM = [(2, 64, 128), (3, 70, 101), (1, 21, 37), (2, 135, 240), (5, 270, 480)]


@pytest.mark.parametrize("case", CASES_STEM)
def test_f16_form_stem_7x7_stride_2(ctx, lib16, case):
    """ResNet-50's conv1 (7x7, stride 2, pad 3, 3 (+1 zero) -> 64 channels, ReLU) on conv_split_stem7_kernel: the float32 yardstick as for
    every other fp16-form layer, per sample and per output channel, at magnitudes from 1e-30 to 1e7 and with every sample at its own."""
    n, h, w = case
    rng = np.random.default_rng(h * w)
    wt = (rng.standard_normal((64, 4, 7, 7)) / np.sqrt(147)).astype(np.float32)
    wt[:, 3] = 0
    wt *= np.exp(2 * rng.standard_normal((64, 1, 1, 1))).astype(np.float32)
    for mag in [np.full(n, m) for m in (1.0, 1e-30, 1e7)] + [10.0 ** np.linspace(-6, 4, n)]:
        x = (rng.standard_normal((n, h, w, 4)) * np.exp(rng.standard_normal((n, h, w, 4))) * mag.reshape(-1, 1, 1, 1)).astype(np.float32)
        x[..., 3] = 0
        b = (rng.standard_normal(64) * np.abs(x).mean()).astype(np.float32)
        pre = conv64(x, wt, b, 3, 2)
        # errors are measured against the range of the PRE-activation values of the (sample, channel): a channel that ReLU leaves
        # almost empty would otherwise be judged relative to its last surviving element
        scale = np.abs(pre).reshape(n, -1, 64).max(1).reshape(n, 1, 1, 64) + 1e-300
        for relu in (L.PP_RELU_LAST, L.PP_RELU_NONE):
            ref = np.maximum(pre, 0) if relu == L.PP_RELU_LAST else pre
            exact, split = both(lib16, lambda: hip_conv_op(ctx, x, wt, b, stride=2, pad=(3, 3), relu=relu))
            assert split.shape == ref.shape and np.isfinite(split).all() and not np.array_equal(exact, split)
            rms = lambda y: float(np.sqrt(np.mean(((y - ref) / scale) ** 2)))
            assert rms(split) <= 1.25 * rms(exact) + 1e-9, (mag[0], relu, rms(split), rms(exact))
            m_s, m_e = float(np.abs((split - ref) / scale).max()), float(np.abs((exact - ref) / scale).max())
            assert m_s <= 1.5 * m_e + 1e-7, (mag[0], relu, m_s, m_e)


def test_kernel_selection_does_not_depend_on_the_batch_capacity(ctx, lib16):
    """Which layers run on which split form is decided from per-SAMPLE quantities only (channels, map size), never from the net's batch
    capacity: nets of the same program created for 1 and for 5 samples select the same kernels, so a rank that holds a shard of a clip
    computes the same bits as the process that holds all of it (tests/test_gpu_sharded.py).  Programs: the detector's image program
    (stem kernel, product kernel from 64 channels, one-column rule, stride-2 forms) and HRNet-W48."""
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    spec = hrnet.HRNetSpec(48, 17, 384, 288)
    progs = [fr.build_image_program(det_sd, 320, 544), hrnet.build_hrnet_program(spec, synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1))]
    for prog in progs:
        kinds = [Net(ctx, prog, max_batch=b, numerics="split_f16").conv_kinds() for b in (1, 5)]
        assert np.array_equal(kinds[0], kinds[1])
        assert (kinds[0] == 2).sum() >= 50


def test_lane_count_does_not_change_results(ctx, lib16):
    """pp_net_set_lane_count (ABI 10): the op -> stream plan of an existing program re-made for 1, 2, 4 and 7 lanes -- HRNet's branches
    spread differently each time, every dependency (RAW on inputs / residuals, WAR / WAW on recycled buffers) an event wait -- same bits,
    in the exact and in the default numerics, and a captured graph is dropped rather than replayed with the old plan"""
    spec = hrnet.HRNetSpec(32, 17, 128, 96)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    rng = np.random.default_rng(31)
    x = np.zeros((5, 128, 96, 4), np.float32)
    x[..., :3] = rng.standard_normal((5, 128, 96, 3))
    for numerics in ("exact", "split"):
        net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=5, numerics=numerics)
        ref = net.forward(x)
        for lanes in (1, 2, 7, 4):
            net.set_lane_count(lanes)
            assert np.array_equal(net.forward(x), ref), (numerics, lanes)
        net.capture(5)
        assert np.array_equal(net.forward(x), ref)
        net.set_lane_count(2)                       # drops the graph
        assert np.array_equal(net.forward(x), ref)
        with pytest.raises(L.PosePipeHipError):
            net.set_lane_count(0)
        net.close()
