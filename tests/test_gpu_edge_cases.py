"""Edge cases the reference path meets in practice: empty / ragged inputs, boxes off the image, degenerate
boxes, maximum sizes, error reporting through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle import boxes as obox
from oracle import preprocess as opre
from posepipeline_amd import _lib as L
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.program import Net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_td(ctx):
    spec = hrnet.HRNetSpec(32, 17, 64, 64)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=8, numerics="exact")      # (module scope: see test_gpu_detector.setup)
    return net, ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="unbiased")


def test_empty_inputs(ctx, small_td):
    net, td = small_td
    frames = np.zeros((1, 48, 64, 3), np.uint8)
    kp, valid = td.run(frames, np.zeros(0, np.int32), np.zeros((0, 4)))
    assert kp.shape == (0, 17, 3) and valid.shape == (0,)
    assert ops.nms(ctx, np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).tolist() == []
    assert ops.nms(ctx, np.array([[0, 0, 5, 5]], np.float32), np.array([0.3], np.float32), 0.5).tolist() == [0]
    kp, _ = ops.flip_merge_decode(ctx, np.zeros((0, 17, 8, 8), np.float32), None, np.zeros((0, 4), np.float32))
    assert kp.shape == (0, 17, 3)


def test_ragged_batch_all_absent_and_capacity(ctx, small_td):
    net, td = small_td
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (2, 48, 64, 3)).astype(np.uint8)
    bb = np.full((3, 4), np.nan)
    kp, valid = td.run(frames, np.array([0, 1, 1], np.int32), bb)
    assert not kp.any() and valid.tolist() == [0, 0, 0]
    with pytest.raises(L.PosePipeHipError, match="exceeds capacity"):
        td.run(frames, np.zeros(5, np.int32), np.tile([1.0, 1, 10, 10], (5, 1)))     # max_batch 8 / 2 flips = 4
    with pytest.raises(L.PosePipeHipError, match="out of range"):
        td.run(frames, np.array([2], np.int32), np.array([[1.0, 1, 10, 10]]))          # frame index past the batch


def test_boxes_outside_and_degenerate(ctx):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (1, 60, 80, 3)).astype(np.uint8)
    boxes = np.array([
        [500.0, 500.0, 40.0, 90.0],      # entirely outside the frame -> all-border crop (zeros)
        [-300.0, -300.0, 50.0, 50.0],
        [10.0, 10.0, 0.0, 0.0],          # zero-size box: singular affine system
        [0.0, 0.0, 1e5, 1e5],            # absurdly large box
        [20.5, 7.25, 1.0, 1.0],          # one-pixel box: extreme magnification
    ])
    res = ops.crop_affine_normalize(ctx, frames, np.zeros(5, np.int32), boxes, out_wh=(48, 64), flip=False, want_crop_u8=True)
    for i, bb in enumerate(boxes):
        _, c, s, crop = opre.top_down_input(frames[0][:, :, ::-1], bb, (48, 64))
        assert np.array_equal(res["crop_u8"][i], crop), i
        assert np.array_equal(res["center_scale"][i], np.concatenate([c, s]))
    assert not res["crop_u8"][0].any() and not res["crop_u8"][1].any()


def test_nms_max_size_and_identical_boxes(ctx):
    rng = np.random.default_rng(2)
    n = 8192
    ctr = rng.uniform(0, 3000, (n, 2))
    wh = rng.uniform(20, 120, (n, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    boxes[100:110] = boxes[100]                 # identical boxes
    scores[100:110] = scores[100]               # ... with identical scores: the lower index wins
    keep = ops.nms(ctx, boxes, scores, 0.5, convention=0)
    assert keep.tolist() == obox.nms_mmcv(boxes, scores, 0.5)
    assert set(range(100, 110)) & set(keep.tolist()) <= {100}       # only the first of the identical boxes may survive
    with pytest.raises(L.PosePipeHipError, match="not in"):
        ops.nms(ctx, np.zeros((8193, 4), np.float32), np.zeros(8193, np.float32), 0.5)


def test_error_strings_and_bad_programs(ctx):
    lib = ctx.lib
    h = C.c_void_p()
    op = L.pp_op(type=L.PP_OP_CONV, in_=0, out=1, res1=-1, res2=-1, cin=4, cout=8, kh=3, kw=3, stride=1, pad_h=1, pad_w=1,
                 dil_h=1, dil_w=1, relu=0, up_log2=0, out_nchw=0, res1_shift=0, res1_off_w=0, w_off=0, b_off=0)
    bufs = (L.pp_buf * 2)(L.pp_buf(8, 8, 4), L.pp_buf(7, 8, 8))           # wrong output height
    blob = np.zeros(4096, np.float32)
    rc = lib.pp_net_create(ctx.handle, (L.pp_op * 1)(op), 1, bufs, 2, L.ptr(blob), blob.size, 1, C.byref(h))
    assert rc == -1 and "does not match out buffer" in L.last_error()
    bufs = (L.pp_buf * 2)(L.pp_buf(8, 8, 4), L.pp_buf(8, 8, 8))
    rc = lib.pp_net_create(ctx.handle, (L.pp_op * 1)(op), 1, bufs, 2, L.ptr(blob), 16, 1, C.byref(h))
    assert rc == -1 and "out of blob" in L.last_error()
    bad = L.pp_op(type=L.PP_OP_CONV, in_=0, out=1, res1=-1, res2=-1, cin=3, cout=8, kh=1, kw=1, stride=1, pad_h=0, pad_w=0,
                  dil_h=1, dil_w=1, relu=0, up_log2=0, out_nchw=0, res1_shift=0, res1_off_w=0, w_off=0, b_off=0)
    x = np.zeros((1, 4, 4, 3), np.float32)
    y = np.zeros((1, 4, 4, 8), np.float32)
    rc = lib.pp_conv2d(ctx.handle, C.byref(bad), 1, 4, 4, L.ptr(x), L.ptr(blob), L.ptr(blob), None, None, L.ptr(y), 0, 0, 0)
    assert rc == -1 and "multiple of 4" in L.last_error()


def test_conv_batch_split_for_large_inputs(tmp_path):
    """Inputs of more than 4 GiB are cut into image ranges (one raw buffer descriptor per launch).  The threshold is
    lowered to 1 MiB in a subprocess (POSEPIPE_CONV_MAX_MB) so that a 7-image batch takes the split path, with shifted
    FPN-style residual, second residual and upsample epilogues; results must equal the oracle."""
    import os
    import subprocess
    import sys
    script = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from posepipeline_amd import _lib
from tests.helpers import hip_conv_op, ref_conv_op
ctx = _lib.Context(0)
rng = np.random.default_rng(21)
x = rng.standard_normal((7, 48, 40, 32)).astype(np.float32)          # 245 KB per image -> 4 images per launch
w = (rng.standard_normal((24, 32, 3, 3)) * 0.1).astype(np.float32)
b = rng.standard_normal(24).astype(np.float32)
cases = [dict(pad=(1, 1), relu=1, res1=rng.standard_normal((7, 48, 40, 24)).astype(np.float32),
              res2=rng.standard_normal((7, 48, 40, 24)).astype(np.float32)),
         dict(pad=(1, 1), up_log2=1, res1=rng.standard_normal((7, 96, 80, 24)).astype(np.float32)),
         dict(pad=(1, 1), res1=rng.standard_normal((7, 24, 20, 24)).astype(np.float32), res1_shift=1),
         dict(pad=(1, 1), stride=2, out_nchw=True)]
for kw in cases:
    got = hip_conv_op(ctx, x, w, b, **kw)
    ref = ref_conv_op(x, w, b, **kw)
    assert np.array_equal(got, ref), kw.keys()
print("SPLIT_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, POSEPIPE_CONV_MAX_MB="1")
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert "SPLIT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_absent_person_rows_are_zero_in_device_outputs_too(ctx, small_td):
    """wrappers/mmpose.py:67-69: a NaN box yields zeros((K, 3)).  With a DEVICE output buffer (PP_MEM_DEVICE) the library has
    to zero those rows itself -- the decode of an all-zero crop is not zero."""
    net, td = small_td
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (2, 48, 64, 3)).astype(np.uint8)
    boxes = np.array([[5.0, 4.0, 30.0, 40.0], [np.nan] * 4, [20.0, 2.0, 25.0, 44.0]], np.float64)
    fidx = np.array([0, 1, 1], np.int32)
    host, valid = td.run(frames, fidx, boxes)
    assert valid.tolist() == [1, 0, 1] and not host[1].any() and host[0].any() and host[2].any()
    dk = ctx.malloc(3 * 17 * 3 * 4)
    ctx.h2d(dk, np.full((3, 17, 3), 7.0, np.float32))
    v = np.zeros(3, np.int32)
    L.check(ctx.lib.pp_topdown_run(td.handle, L.ptr(frames), 2, 48, 64, L.PP_MEM_HOST, L.ptr(fidx), L.ptr(boxes), 3,
                                   C.c_void_p(dk), L.PP_MEM_DEVICE, L.ptr(v)), "pp_topdown_run")
    dev = np.empty((3, 17, 3), np.float32)
    ctx.d2h(dev, dk)
    ctx.free(dk)
    assert np.array_equal(dev, host)
