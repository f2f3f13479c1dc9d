"""The detector's per-frame decision margins (pp_detector_enable_margins, ABI 10; det_post.hip / nms.hip) against a numpy restatement
of their DEFINITIONS evaluated on the device's own intermediate values (exact numerics: those equal the oracle's bit for bit, so the
restatement below runs on oracle/detector.py's functions)."""
import numpy as np
import pytest

from oracle import boxes as obox
from oracle import detector as odet
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import synth
from tests.test_gpu_detector import synth_frame

pytestmark = pytest.mark.gpu
f32 = np.float32
W = 1.25          # score_weight


def iou_matrix(b):
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.maximum(np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]), 0)
    h = np.maximum(np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]), 0)
    inter = (w * h).astype(f32)
    union = ((area[:, None] + area[None, :]).astype(f32) - inter).astype(f32)
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0).astype(f32)


def nms_margin(boxes, scores, thr, w):
    """the definition in nms.hip: walk the boxes in score order (ties: lower index first); a kept box contributes min over its kept
    predecessors of (thr - IoU), a suppressed one the max over its kept suppressors of min(IoU - thr, w * score lead); the frame's
    figure is the minimum over the boxes (clamped at 0: the division and mmcv's product form can disagree by an ulp at the threshold)"""
    n = len(boxes)
    if n == 0:
        return np.inf
    order = np.argsort(-scores, kind="stable")
    b, s = boxes[order].astype(f32), scores[order].astype(f32)
    pos_of = np.empty(n, np.int64)
    pos_of[order] = np.arange(n)
    kept = np.zeros(n, bool)
    kept[pos_of[np.array(obox.nms_mmcv(boxes, scores, thr), np.int64)]] = True
    d = iou_matrix(b) - f32(thr)                                   # d[i, j]
    pred = kept[:, None] & (np.arange(n)[:, None] < np.arange(n)[None, :])
    m_keep = np.where(pred, -d, np.inf).min(axis=0)
    lead = (f32(w) * (s[:, None] - s[None, :])).astype(f32)
    m_sup = np.where(pred & (d > 0), np.minimum(d, lead), -1.0).max(axis=0)
    mj = np.where(kept, np.maximum(m_keep, 0.0), np.maximum(m_sup, 0.0))
    return float(mj.min())


def test_margins_equal_their_definitions(ctx):
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    for k, g in (("detector.rpn_head.rpn_cls.weight", 0.5), ("detector.rpn_head.rpn_reg.weight", 0.1),
                 ("detector.roi_head.bbox_head.fc_reg.weight", 0.2)):
        sd[k] = (sd[k] * g).astype(np.float32)
    rng = np.random.default_rng(4)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(2)])
    det = fr.Detector(ctx, sd, 135, 240, max_frames=2, numerics="exact")
    with pytest.raises(Exception, match="not enabled"):
        det.margins(2)
    det.enable_margins(True, W)
    dets, props = det.run(frames, want_proposals=True)
    m = det.margins(2)
    assert m.shape == (2, 8) and (m >= 0).all()
    names = fr.Detector.MARGIN_NAMES
    model = odet.FasterRCNNRef(sd)
    for f in range(2):
        ref, mid = odet.detect(model, frames[f][:, :, ::-1], want_intermediates=True)
        assert np.array_equal(dets[f], ref) and np.array_equal(props[f], mid["proposals"])       # margins on: same outputs
        got = dict(zip(names, m[f]))
        # rpn_cut: gap across the top-1000 cut of every level that has one
        gaps, cand_b, cand_s, cand_l = [], [], [], []
        for lvl, (c, r) in enumerate(zip(mid["cls_maps"], mid["reg_maps"])):
            sc = odet.sigmoid_f32(c[0].reshape(-1))
            order = np.argsort(-sc, kind="stable")
            if len(sc) > 1000:
                gaps.append(float(sc[order[999]] - sc[order[1000]]))
                order = order[:1000]
            anchors = odet.grid_anchors(c.shape[1], c.shape[2], odet.STRIDES[lvl])
            cand_b.append(odet.delta2bbox(anchors[order], r[0].reshape(-1, 4)[order]))
            cand_s.append(sc[order])
            cand_l.append(np.full(len(order), lvl))
        assert got["rpn_cut"] == pytest.approx(min(gaps), rel=1e-6, abs=1e-12)
        # rpn_nms / rpn_top on the level-offset candidates (batched_nms)
        b, s, l = np.concatenate(cand_b), np.concatenate(cand_s), np.concatenate(cand_l)
        valid = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        b, s, l = b[valid], s[valid], l[valid]
        off = (l.astype(f32) * f32(b.max() + f32(1))).astype(f32)
        bn = (b + off[:, None]).astype(f32)
        assert got["rpn_nms"] == pytest.approx(nms_margin(bn, s, 0.7, W), rel=2e-4, abs=2e-7)
        keep = np.array(obox.nms_mmcv(bn, s, 0.7), np.int64)
        want_top = float(s[keep[999]] - s[keep[1000]]) if len(keep) > 1000 else np.inf
        assert got["rpn_top"] == pytest.approx(want_top, rel=1e-6, abs=1e-12)
        # roi_level / score_thr over the proposals
        rois = mid["proposals"]
        scale = np.sqrt(((rois[:, 2] - rois[:, 0]) * (rois[:, 3] - rois[:, 1])).astype(f32)).astype(f32)
        v = np.log2((scale / f32(56)).astype(f32).astype(np.float64) + np.float64(f32(1e-6)))
        assert got["roi_level"] == pytest.approx(float(np.abs(v[:, None] - np.array([1.0, 2.0, 3.0])).min()), rel=1e-5, abs=1e-7)
        sc_roi = odet.softmax_fg(mid["cls"])
        assert got["score_thr"] == pytest.approx(float(np.abs(sc_roi - f32(0.05)).min()), rel=1e-6, abs=1e-9)
        # det_nms / det_top / det_order over the scored boxes
        boxes = odet.delta2bbox(rois, mid["reg"], stds=(0.1, 0.1, 0.2, 0.2))
        boxes = (boxes / mid["scale_factor"][None, :].astype(f32)).astype(f32)
        sel = sc_roi > f32(0.05)
        fb, fs = boxes[sel], sc_roi[sel]
        assert got["det_nms"] == pytest.approx(nms_margin(fb, fs, 0.5, W), rel=2e-4, abs=2e-7)
        keep2 = np.array(obox.nms_mmcv(fb, fs, 0.5), np.int64)
        want_top2 = float(fs[keep2[99]] - fs[keep2[100]]) if len(keep2) > 100 else np.inf
        assert got["det_top"] == pytest.approx(want_top2, rel=1e-6, abs=1e-12)
        out_s = ref[:, 4]
        want_order = float(np.min(out_s[:-1] - out_s[1:])) if len(out_s) > 1 else np.inf
        assert got["det_order"] == pytest.approx(want_order, rel=1e-6, abs=1e-12)
    # margins off again: the call is refused, the detections unchanged
    det.enable_margins(False, W)
    assert all(np.array_equal(a, b) for a, b in zip(det.run(frames), dets))
    with pytest.raises(Exception, match="not enabled"):
        det.margins(2)
