"""DeepSortYOLOv4 path on the CPU: oracle vs the golden vectors generated from the reference (real PIL letterbox,
extract_image_patch), and the product's host-side tables / program builders vs the oracle."""
import os

import numpy as np

from oracle import reid as oreid
from oracle import yolo as oyolo
from posepipeline_amd.models import mars, yolov4

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_letterbox_matches_pil_golden():
    g = np.load(os.path.join(GOLD, "letterbox.npz"))
    for i in range(int(g["n"])):
        out, _ = oyolo.letterbox(g[f"in{i}"], tuple(int(v) for v in g[f"size{i}"]))
        assert np.array_equal(out, g[f"out{i}"]), i          # Pillow's 8-bit bicubic, bit for bit


def test_patch_rect_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "reid_patch.npz"))
    hw = tuple(int(v) for v in g["image_hw"])
    for b, is_int, rect in zip(g["boxes"], g["is_int"], g["rects"]):
        bb = b.astype(np.int64) if is_int else b
        for fn in (oreid.patch_rect, mars.patch_rect):         # oracle and product host code
            got = fn(bb, hw)
            assert ([-1] * 4 if got is None else list(got)) == list(rect)


def test_product_bicubic_tables_equal_oracle():
    for src, dst in ((1920, 416), (1080, 234), (240, 96), (40, 96), (77, 77)):
        tab, ksize = yolov4.pil_bicubic_table(src, dst)
        ref = oyolo.pil_coeffs(src, dst)
        assert tab.shape == (dst, 2 + ksize)
        for xx, (xmin, cnt, k) in enumerate(ref):
            assert tab[xx, 0] == xmin and tab[xx, 1] == cnt and cnt <= ksize
            assert np.array_equal(tab[xx, 2:2 + cnt], k) and not tab[xx, 2 + cnt:].any()
    assert yolov4.letterbox_geometry(1080, 1920) == (416, 234)


def test_yolov4_program_shape():
    layers = yolov4._layers()
    assert len(layers) == 110                                  # 72 backbone + 38 head convolutions
    shapes = yolov4.yolov4_param_shapes()
    n_params = sum(int(np.prod(s)) for s in shapes.values())
    assert 64.0e6 < n_params < 64.8e6                          # YOLOv4: 64.4 M parameters (incl. BN statistics)
    sd = yolov4.synth_params(yolov4.yolov4_param_shapes(num_classes=2), seed=0)
    prog = yolov4.build_yolov4_program(sd, size=64, num_classes=2)
    assert {"input", "y19", "y38", "y76"} <= set(prog.named)
    assert [prog.bufs[prog.named[k]] for k in ("y19", "y38", "y76")] == [(2, 2, 21), (4, 4, 21), (8, 8, 21)]
    full = yolov4.build_yolov4_program(yolov4.synth_params(shapes, seed=0), size=416)
    assert 29.0e9 < full.flops / 2 < 31.0e9                    # ~30 GMAC at 416x416 (60 BFLOPs)


def test_mars_program_shape():
    sd = yolov4.synth_params(mars.mars_param_shapes(), seed=1)
    prog = mars.build_mars_program(sd)
    assert prog.bufs[prog.named["input"]] == (128, 64, 4) and prog.bufs[prog.named["features"]] == (1, 1, 128)
    assert mars.tf_same(63, 3, 2) == (1, 0, 32) and mars.tf_same(32, 3, 2) == (0, 1, 16) and mars.tf_same(32, 1, 2) == (0, 0, 16)
    n_params = sum(int(np.prod(s)) for s in mars.mars_param_shapes().values())
    assert 2.7e6 < n_params < 2.9e6                            # the deep_sort paper's 2.8 M parameter network


def test_oracle_yolo_decode_and_nms_known_answers():
    # one confident cell, everything else suppressed: the decoded box must be the anchor box centred on the cell
    nc, g = 2, 2
    out = np.full((1, g, g, 3 * (5 + nc)), -20.0, np.float32)
    f = out.reshape(g, g, 3, 5 + nc)
    f[1, 0, 1, :] = [0.0, 0.0, 0.0, 0.0, 20.0, 20.0, -20.0]    # cell (y=1, x=0), anchor 1, class 0
    anchors = np.array([[10, 12], [16, 24], [30, 20]], np.float32)
    boxes, scores = oyolo.boxes_and_scores(out, anchors, nc, (64, 64), (64, 64))
    i = (1 * g + 0) * 3 + 1
    assert scores[i, 0] > 0.99 and (np.delete(scores[:, 0], i) < 1e-6).all()
    assert np.allclose(boxes[i], [48 - 12, 16 - 8, 48 + 12, 16 + 8], atol=1e-4)      # y1, x1, y2, x2
    b = np.array([[0, 0, 10, 10], [0, 1, 10, 11], [20, 20, 30, 30], [0, 0, 10, 10]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.9], np.float32)
    assert list(oyolo.tf_nms(b, s, 200, 0.5)) == [0, 2]       # tie 0/3 -> lower index; 1 overlaps 0 (IoU .82)
    assert list(oyolo.tf_nms(b, s, 1, 0.5)) == [0]


# ---- independent torch-CPU statements of the two networks (unfolded BatchNorm, torch's own ops and summation order) -----
def _torch_yolov4(sd, x_nhwc):
    """yolo4/model.py:78-190 written against torch ops only (NCHW, F.mish / F.leaky_relu / torch.cat / F.interpolate)."""
    import torch
    import torch.nn.functional as F
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    idx = [0]

    def conv(x, stride=1, act="mish"):
        i = idx[0]
        idx[0] += 1
        w = t[f"l{i}.weight"]
        k = w.shape[2]
        if stride == 2:                    # ZeroPadding2D(((1, 0), (1, 0))) + 'valid'
            y = F.conv2d(F.pad(x, (1, 0, 1, 0)), w, t.get(f"l{i}.bias"), 2, 0)
        else:
            y = F.conv2d(x, w, t.get(f"l{i}.bias"), 1, k // 2)
        if f"l{i}.bn.gamma" in t:
            y = F.batch_norm(y, t[f"l{i}.bn.mean"], t[f"l{i}.bn.var"], t[f"l{i}.bn.gamma"], t[f"l{i}.bn.beta"], False, 0.1, 1e-3)
        return {"mish": F.mish, "leaky": lambda v: F.leaky_relu(v, 0.1), None: lambda v: v}[act](y)

    def resblock(x, blocks):
        pre = conv(x, 2)
        short = conv(pre)
        main = conv(pre)
        for _ in range(blocks):
            main = main + conv(conv(main))
        return conv(torch.cat([conv(main), short], 1))

    def five(x):
        for _ in range(5):
            x = conv(x, act="leaky")
        return x

    x = torch.from_numpy(np.ascontiguousarray(np.transpose(x_nhwc, (0, 3, 1, 2))))
    x = conv(x)
    x = resblock(x, 1)
    x = resblock(x, 2)
    f76 = x = resblock(x, 8)
    f38 = x = resblock(x, 8)
    x = resblock(x, 4)
    y19 = conv(conv(conv(x, act="leaky"), act="leaky"), act="leaky")
    y19 = torch.cat([F.max_pool2d(y19, k, 1, k // 2) for k in (13, 9, 5)] + [y19], 1)
    y19 = conv(conv(conv(y19, act="leaky"), act="leaky"), act="leaky")
    up = F.interpolate(conv(y19, act="leaky"), scale_factor=2, mode="nearest")
    y38 = five(torch.cat([conv(f38, act="leaky"), up], 1))
    up = F.interpolate(conv(y38, act="leaky"), scale_factor=2, mode="nearest")
    y76 = five(torch.cat([conv(f76, act="leaky"), up], 1))
    o76 = conv(conv(y76, act="leaky"), act=None)
    y38 = five(torch.cat([conv(y76, 2, act="leaky"), y38], 1))
    o38 = conv(conv(y38, act="leaky"), act=None)
    y19 = five(torch.cat([conv(y38, 2, act="leaky"), y19], 1))
    o19 = conv(conv(y19, act="leaky"), act=None)
    return [np.transpose(o.numpy(), (0, 2, 3, 1)) for o in (o19, o38, o76)]


def test_yolov4_oracle_vs_torch():
    nc = 2
    sd = yolov4.synth_params(yolov4.yolov4_param_shapes(nc), seed=8)
    x = np.random.default_rng(8).uniform(0, 1, (1, 64, 64, 3)).astype(np.float32)
    ref = _torch_yolov4(sd, x)
    got = oyolo.YOLOv4Ref(sd, nc).forward(x)
    for g, r in zip(got, ref):
        assert g.shape == r.shape
        np.testing.assert_allclose(g, r, rtol=2e-3, atol=2e-4)


def test_mars_oracle_vs_torch():
    """tools/freeze_model.py:119-229 against torch ops (NCHW, explicit TensorFlow-SAME padding, F.elu, unfolded BN)."""
    import torch
    import torch.nn.functional as F
    sd = yolov4.synth_params(mars.mars_param_shapes(), seed=9)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}

    def bn(x, p):
        return F.batch_norm(x, t[p + ".mean"], t[p + ".var"], None, t[p + ".beta"], False, 0.1, 1e-3)

    def same(x, k, s):
        pads = []
        for n in (x.shape[3], x.shape[2]):          # F.pad order: W first
            tot = max((-(-n // s) - 1) * s + k - n, 0)
            pads += [tot // 2, tot - tot // 2]
        return F.pad(x, pads)

    def conv(x, name, stride=1, norm=True, act=True):
        w = t[name + ".weight"]
        y = F.conv2d(same(x, w.shape[2], stride), w, t.get(name + ".bias"), stride, 0)
        if norm:
            y = bn(y, name + ".bn")
        return F.elu(y) if act else y

    rng = np.random.default_rng(9)
    patches = rng.integers(0, 256, (2, 128, 64, 3), dtype=np.uint8)
    x = torch.from_numpy(np.ascontiguousarray(np.transpose(patches[..., ::-1].astype(np.float32), (0, 3, 1, 2))))
    x = conv(conv(x, "conv1_1"), "conv1_2")
    x = F.max_pool2d(x, 3, 2, 0)
    for scope, c, inc, first in mars.BLOCKS:
        net = x if first else F.elu(bn(x, scope + ".bn"))
        y = conv(conv(net, scope + ".1", 2 if inc else 1), scope + ".2", norm=False, act=False)
        x = (conv(x, scope + ".projection", 2, norm=False, act=False) if inc else x) + y
    flat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)            # slim.flatten of an NHWC tensor
    f = F.elu(F.batch_norm(flat @ t["fc1.weight"].T, t["fc1.bn.mean"], t["fc1.bn.var"], None, t["fc1.bn.beta"], False, 0.1, 1e-3))
    f = F.batch_norm(f, t["ball.mean"], t["ball.var"], None, t["ball.beta"], False, 0.1, 1e-3)
    ref = (f / torch.sqrt(1e-8 + (f * f).sum(1, keepdim=True))).numpy()
    got = oreid.MarsSmall128Ref(sd).forward(patches)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5)
