"""Pin the detector oracle's network wiring (oracle/detector.py FasterRCNNRef) against an independent torch-CPU
statement of mmdet's ResNet-50 (style='pytorch') + FPN + RPNHead + Shared2FCBBoxHead: torch ops, UNFOLDED eval-mode
batch_norm, torch.flatten of an NCHW tensor for the FC layers.  Agreement is to float32 round-off.
Also: RoIAlign (aligned, adaptive sampling) against a deliberately naive per-sample loop."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import detector as odet
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import synth


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def test_faster_rcnn_networks_vs_torch():
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    t = _t(sd)
    P = "detector."

    def cb(x, conv, bn, stride=1, pad=0):
        y = F.conv2d(x, t[P + conv + ".weight"], None, stride, pad)
        return F.batch_norm(y, t[P + bn + ".running_mean"], t[P + bn + ".running_var"], t[P + bn + ".weight"], t[P + bn + ".bias"],
                            False, 0.1, 1e-5)

    def conv(x, name, stride=1, pad=0):
        return F.conv2d(x, t[P + name + ".weight"], t[P + name + ".bias"], stride, pad)

    rng = np.random.default_rng(3)
    x_nhwc = rng.standard_normal((1, 64, 96, 3)).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(np.transpose(x_nhwc, (0, 3, 1, 2))))
    # backbone
    y = F.max_pool2d(F.relu(cb(x, "backbone.conv1", "backbone.bn1", 2, 3)), 3, 2, 1)
    feats = []
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2))):
        for b in range(blocks):
            q = f"backbone.layer{li + 1}.{b}."
            s = stride if b == 0 else 1
            idn = cb(y, q + "downsample.0", q + "downsample.1", s, 0) if b == 0 else y
            z = F.relu(cb(y, q + "conv1", q + "bn1"))
            z = F.relu(cb(z, q + "conv2", q + "bn2", s, 1))
            y = F.relu(cb(z, q + "conv3", q + "bn3") + idn)
        feats.append(y)
    # FPN (mmdet: F.interpolate(nearest) top-down, 3x3 output convs, extra level by max_pool2d(1, stride=2))
    lat = [conv(f, f"neck.lateral_convs.{i}.conv") for i, f in enumerate(feats)]
    for i in range(3, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    outs = [conv(l, f"neck.fpn_convs.{i}.conv", 1, 1) for i, l in enumerate(lat)]
    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    cls_t, reg_t = [], []
    for f in outs:
        h = F.relu(conv(f, "rpn_head.rpn_conv", 1, 1))
        cls_t.append(conv(h, "rpn_head.rpn_cls"))
        reg_t.append(conv(h, "rpn_head.rpn_reg"))

    model = odet.FasterRCNNRef({k: v for k, v in sd.items()})
    ref_feats = model.backbone(x_nhwc.copy() if x_nhwc.shape[3] == 3 else x_nhwc)
    ref_outs = model.fpn(ref_feats)
    ref_cls, ref_reg = model.rpn_head(ref_outs)
    nchw = lambda a: np.transpose(a, (0, 3, 1, 2))
    for a, b in zip(ref_outs, outs):
        np.testing.assert_allclose(nchw(a), b.numpy(), rtol=2e-3, atol=2e-4)
    for a, b in zip(ref_cls + ref_reg, cls_t + reg_t):
        np.testing.assert_allclose(nchw(a), b.numpy(), rtol=2e-3, atol=2e-4)

    # RoI head: torch flattens NCHW RoI features (c, h, w) before the first FC
    roi = rng.standard_normal((5, 7, 7, 256)).astype(np.float32)
    flat = torch.from_numpy(np.ascontiguousarray(np.transpose(roi, (0, 3, 1, 2)))).flatten(1)
    B = P + "roi_head.bbox_head."
    h = F.relu(F.linear(flat, t[B + "shared_fcs.0.weight"], t[B + "shared_fcs.0.bias"]))
    h = F.relu(F.linear(h, t[B + "shared_fcs.1.weight"], t[B + "shared_fcs.1.bias"]))
    cls, reg = F.linear(h, t[B + "fc_cls.weight"], t[B + "fc_cls.bias"]), F.linear(h, t[B + "fc_reg.weight"], t[B + "fc_reg.bias"])
    rc, rr = model.roi_head(roi)
    np.testing.assert_allclose(rc, cls.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(rr, reg.numpy(), rtol=2e-3, atol=2e-4)


def test_roi_align_vs_naive_loop():
    """mmcv RoIAlign(output 7, sampling_ratio 0, aligned=True, 'avg'): every bin averages ceil(roi/7)^2 bilinear samples
    taken at (start - 0.5) + (i + 0.5) * bin / grid."""
    rng = np.random.default_rng(4)
    feat = rng.standard_normal((20, 30, 8)).astype(np.float32)
    for roi in ([3.2, 4.1, 17.9, 15.3], [0.0, 0.0, 29.0, 19.0], [-5.0, -3.0, 8.0, 6.0], [10.0, 10.0, 10.5, 10.5]):
        roi = np.array(roi, np.float32)
        scale = 0.5
        got = odet.roi_align(feat, roi, scale)
        x1, y1, x2, y2 = (float(v) * scale - 0.5 for v in roi)
        rw, rh = x2 - x1, y2 - y1
        bw, bh = rw / 7, rh / 7
        gw, gh = int(np.ceil(rw / 7)), int(np.ceil(rh / 7))
        H, W = feat.shape[:2]
        ref = np.zeros((7, 7, 8))
        for py in range(7):
            for px in range(7):
                acc = np.zeros(8)
                for iy in range(gh):
                    for ix in range(gw):
                        y = y1 + py * bh + (iy + 0.5) * bh / gh
                        x = x1 + px * bw + (ix + 0.5) * bw / gw
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        y, x = max(y, 0.0), max(x, 0.0)
                        y0, x0 = int(y), int(x)
                        if y0 >= H - 1:
                            y0 = ya = H - 1
                            y = float(y0)
                        else:
                            ya = y0 + 1
                        if x0 >= W - 1:
                            x0 = xa = W - 1
                            x = float(x0)
                        else:
                            xa = x0 + 1
                        ly, lx = y - y0, x - x0
                        acc += ((1 - ly) * (1 - lx) * feat[y0, x0] + (1 - ly) * lx * feat[y0, xa] + ly * (1 - lx) * feat[ya, x0] +
                                ly * lx * feat[ya, xa])
                ref[py, px] = acc / max(gh * gw, 1)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


# ---- margin-aware (three-valued) restatement of the discrete decisions: oracle/detector_margins.py -------------------------
def test_three_valued_nms_brackets_every_nearby_evaluation():
    """nms3 / _topk3: with eps = 0 they reproduce mmcv's greedy NMS; with eps > 0, every evaluation whose scores and boxes are
    within a tenth of eps of this one keeps all certain boxes and nothing outside the possible ones"""
    from oracle import boxes as obox
    from oracle import detector_margins as odm
    rng = np.random.default_rng(31)
    n_uncertain = 0
    for trial in range(6):
        n = 300
        c = rng.uniform(0, 400, (12, 2))
        ctr = c[rng.integers(0, 12, n)] + rng.normal(0, 10, (n, 2))
        wh = rng.uniform(30, 120, (n, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        scores = np.round(rng.uniform(0, 1, n), 3).astype(np.float32)          # many near-ties at the 1e-3 level
        plain = set(obox.nms_mmcv(boxes, scores, 0.5))
        st0 = odm.nms3(boxes, scores, np.full(n, odm.K), 0.5, 0.0, 0.0)
        assert {i for i in np.flatnonzero(st0 == odm.K)} <= plain and plain <= {i for i in np.flatnonzero(st0 != odm.S)}
        eps = 2e-3
        st = odm.nms3(boxes, scores, np.full(n, odm.K), 0.5, eps, eps)
        certain, possible = set(np.flatnonzero(st == odm.K)), set(np.flatnonzero(st != odm.S))
        assert certain <= plain <= possible
        n_uncertain += len(possible) - len(certain)
        for k in range(5):
            s2 = (scores + rng.uniform(-eps / 10, eps / 10, n)).astype(np.float32)
            b2 = (boxes + rng.uniform(-1e-3, 1e-3, boxes.shape)).astype(np.float32)
            near = set(obox.nms_mmcv(b2, s2, 0.5))
            assert certain <= near <= possible, (trial, k, certain - near, near - possible)
        # top-k in three-valued logic
        kept = np.where(st != odm.S)[0]
        tk = odm._topk3(scores, st, 20, eps)
        order = [i for i in np.argsort(-scores, kind="stable") if i in plain][:20]
        assert set(np.flatnonzero(tk == odm.K)) <= set(order) <= set(np.flatnonzero(tk != odm.S))
    assert n_uncertain > 0            # the near-ties of these cases do make some decisions uncertain


def test_check_between_matches_rows():
    from oracle import detector_margins as odm
    certain = np.array([[0, 0, 10, 10, 0.9], [20, 20, 40, 50, 0.6]], np.float32)
    possible = np.concatenate([certain, np.array([[5, 5, 9, 9, 0.3]], np.float32)])
    dev = certain + np.array([1e-3, -1e-3, 0, 1e-3, 5e-5], np.float32)
    assert odm.check_between(dev, certain, possible, 1e-2)[:2] == ([], [])
    miss, unex, _ = odm.check_between(dev[:1], certain, possible, 1e-2)
    assert len(miss) == 1 and not unex
    miss, unex, _ = odm.check_between(np.concatenate([dev, [[100, 100, 120, 120, 0.2]]]).astype(np.float32), certain, possible, 1e-2)
    assert not miss and len(unex) == 1
    assert odm.check_between(dev, certain, possible, 1e-4)[0]                      # tolerance is enforced
