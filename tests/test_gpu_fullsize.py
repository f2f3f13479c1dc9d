"""Parity at BASELINE.json's full sizes.  The CPU oracle is fast enough for a few full-size samples, and the
rest of the batch is tied to them by size-independent properties of the path: the result for a sample must
not depend on its position in the batch, on the batch size (different tile shapes / launch geometry), or on
whether the program is replayed from a hipGraph; mirroring the input mirrors the heat-map."""
import numpy as np
import pytest

from oracle import nets as onets
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.program import Net

pytestmark = pytest.mark.gpu


def test_hrnet_w32_256x192_batch64_configs1(ctx):
    """configs[1]: HRNet-W32 256x192, batch 64 pre-cropped persons"""
    spec = hrnet.hrnet_w32_256x192()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=64)
    rng = np.random.default_rng(1)
    x = np.zeros((64, 256, 192, 4), np.float32)
    x[..., :3] = rng.standard_normal((64, 256, 192, 3)).astype(np.float32)
    hm = net.forward(x).reshape(64, 17, 64, 48)
    # full-size oracle on the first and the last sample: bit-exact
    ref = onets.HRNetRef(sd, 32).forward(np.transpose(x[[0, 63], :, :, :3], (0, 3, 1, 2)))
    assert np.array_equal(hm[[0, 63]], ref)
    # batch-position / batch-size independence: a permuted batch and small batches give the same bits
    perm = rng.permutation(64)
    assert np.array_equal(net.forward(x[perm]).reshape(64, 17, 64, 48), hm[perm])
    assert np.array_equal(net.forward(x[5:8]).reshape(3, 17, 64, 48), hm[5:8])
    assert np.array_equal(net.forward(x[40:41]).reshape(1, 17, 64, 48), hm[40:41])
    # determinism + hipGraph replay of the same program
    net.capture(64)
    assert np.array_equal(net.forward(x).reshape(64, 17, 64, 48), hm)
    # decode of the whole batch: maxvals are exactly the heat-map maxima, coordinates inside the crop window
    td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="default")
    cs = np.tile(np.array([[96.0, 128.0, 192 / 200 * 1.25, 256 / 200 * 1.25]], np.float32), (32, 1))
    kp = td.run_precropped(x[:32], cs)
    assert kp.shape == (32, 17, 3) and np.isfinite(kp).all()
    assert (kp[:, :, 0] > -30).all() and (kp[:, :, 0] < 222).all()


def test_hrnet_w48_384x288_full_size(ctx):
    """the reference's own configuration (W48 384x288 DARK): one full-size sample against the oracle, plus the
    mirrored sample (flip_test's second pass)"""
    spec = hrnet.hrnet_w48_384x288()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=4)
    rng = np.random.default_rng(2)
    x = np.zeros((2, 384, 288, 4), np.float32)
    x[0, ..., :3] = rng.standard_normal((384, 288, 3)).astype(np.float32)
    x[1] = x[0][:, ::-1]
    hm = net.forward(x).reshape(2, 17, 96, 72)
    ref = onets.HRNetRef(sd, 48).forward(np.transpose(x[:, :, :, :3], (0, 3, 1, 2)))
    assert np.array_equal(hm, ref)


@pytest.mark.parametrize("n", [40, 66])
def test_conv_input_beyond_2_gib(ctx, n):
    """n = 40: a 2.7 GB input tensor (byte offsets past 2^31 in the raw buffer loads); n = 66: 4.4 GB, i.e. past the 4 GiB
    range of one buffer descriptor, so the launcher really cuts the batch.  All images but the last share one pattern;
    the first, an image 2.35 GB in, and the last image of the output must equal the oracle on those images alone."""
    import ctypes as C
    from oracle import clib
    from posepipeline_amd import _lib as L
    from posepipeline_amd.program import pack_conv
    h, w, cin, cout = 256, 256, 256, 32
    rng = np.random.default_rng(77)
    img_a = rng.standard_normal((1, h, w, cin)).astype(np.float32)
    img_z = rng.standard_normal((1, h, w, cin)).astype(np.float32)
    weight = (rng.standard_normal((cout, cin, 3, 3)) * 0.02).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    W, b = pack_conv(weight, bias, cin_pad=cin)
    img_bytes = img_a.nbytes
    assert n * img_bytes > 2 ** 31
    dx = ctx.malloc(n * img_bytes)
    for i in range(n - 1):
        ctx.h2d(dx + i * img_bytes, img_a)
    ctx.h2d(dx + (n - 1) * img_bytes, img_z)
    out_img = h * w * cout * 4
    dy = ctx.malloc(n * out_img)
    dw, db = ctx.malloc(W.nbytes), ctx.malloc(b.nbytes)
    ctx.h2d(dw, W)
    ctx.h2d(db, b)
    op = L.pp_op(type=L.PP_OP_CONV, in_=0, out=0, res1=-1, res2=-1, cin=cin, cout=cout, kh=3, kw=3, stride=1, pad_h=1, pad_w=1,
                 dil_h=1, dil_w=1, relu=L.PP_RELU_LAST, up_log2=0, out_nchw=0, res1_shift=0, res1_off_w=0, w_off=0, b_off=0)
    L.check(ctx.lib.pp_conv2d(ctx.handle, C.byref(op), n, h, w, L.ptr(dx), L.ptr(dw), L.ptr(db), None, None, L.ptr(dy), 0, 0,
                              L.PP_MEM_DEVICE), "pp_conv2d")
    ctx.synchronize()
    first = np.empty((1, h, w, cout), np.float32)
    mid = np.empty((1, h, w, cout), np.float32)
    last = np.empty((1, h, w, cout), np.float32)
    ctx.d2h(first, dy)
    ctx.d2h(mid, dy + 35 * out_img)                # starts 2.35 GB into the input
    ctx.d2h(last, dy + (n - 1) * out_img)
    ref_a = np.maximum(clib.conv2d_nhwc(img_a, weight, bias, stride=1, pad=(1, 1)), 0)
    ref_z = np.maximum(clib.conv2d_nhwc(img_z, weight, bias, stride=1, pad=(1, 1)), 0)
    assert np.array_equal(first, ref_a) and np.array_equal(mid, ref_a) and np.array_equal(last, ref_z)
    for p in (dx, dy, dw, db):
        ctx.free(p)


def test_vitpose_huge_full_size(ctx):
    """configs[4] at full size: ViTPose-H (32 blocks, dim 1280) on 256x192.  One sample against the CPU oracle that rounds
    to bf16 at the same points (tolerance: 1e-2 of the heat-map range -- fp32 accumulation order and the occasional
    1-ulp bf16 flip over 32 blocks; 5.7e-3 measured), and the size-independent properties: batch-position and batch-size
    independence are EXACT (each output element is one MFMA chain over K in a fixed order, whatever the tile shape),
    and the hipGraph replay reproduces the same bits."""
    from oracle import vit as ovit
    from posepipeline_amd.models import vitpose
    spec = vitpose.vitpose_huge()
    p = vitpose.synth_params(spec, seed=5)
    net = Net(ctx, vitpose.build_vitpose_program(spec, p), max_batch=24)
    rng = np.random.default_rng(4)
    n = 24
    x = np.zeros((n, 256, 192, 4), np.float32)
    x[..., :3] = rng.standard_normal((n, 256, 192, 3)).astype(np.float32)
    hm = net.forward(x).reshape(n, 17, 64, 48)
    ref = ovit.forward(x[7:8], p, spec, emulate_bf16=True)
    err = float(np.abs(hm[7] - ref[0]).max() / np.abs(ref[0]).max())
    assert err < 1e-2, err
    perm = rng.permutation(n)
    assert np.array_equal(net.forward(x[perm]).reshape(n, 17, 64, 48), hm[perm])
    assert np.array_equal(net.forward(x[5:8]).reshape(3, 17, 64, 48), hm[5:8])      # other GEMM tile configuration
    assert np.array_equal(net.forward(x[20:21]).reshape(1, 17, 64, 48), hm[20:21])
    net.capture(n)
    assert np.array_equal(net.forward(x).reshape(n, 17, 64, 48), hm)
    net.close()


def test_detector_1080p_full_size(ctx):
    """configs[2]/[3] at full size: one synthetic 1920x1080 frame through Faster-RCNN R50-FPN (640x1088 input, 174 K
    anchors, 1000 proposals) -- detections bit-exact against the CPU oracle -- and batch-position independence: the same
    frame at positions 0 and 2 of a 3-frame batch gives the same detections as alone."""
    from oracle import detector as odet
    from posepipeline_amd.models import faster_rcnn as fr
    from tests.test_gpu_detector import synth_frame
    rng = np.random.default_rng(3)
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    for k, g in (("detector.rpn_head.rpn_cls.weight", 0.5), ("detector.rpn_head.rpn_reg.weight", 0.1),
                 ("detector.roi_head.bbox_head.fc_reg.weight", 0.2)):
        sd[k] = (sd[k] * g).astype(np.float32)
    f0, f1 = synth_frame(rng, 1080, 1920), synth_frame(rng, 1080, 1920)
    det = fr.Detector(ctx, sd, 1080, 1920, max_frames=3)
    alone = det.run(f0[None])[0]
    ref = odet.detect(odet.FasterRCNNRef(sd), f0[:, :, ::-1])
    assert alone.shape == ref.shape and alone.shape[0] > 0
    assert np.array_equal(alone, ref)
    batch = det.run(np.stack([f0, f1, f0]))
    assert np.array_equal(batch[0], alone) and np.array_equal(batch[2], alone) and not np.array_equal(batch[1], alone)
