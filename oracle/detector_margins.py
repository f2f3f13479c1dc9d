"""Margin-aware restatement of the detector's DISCRETE decisions.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

oracle/detector.py follows mmdet's Faster-RCNN test path (3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:101-109,
reached from pose_pipeline/wrappers/mmtrack.py:45) decision by decision: per-level top-1000, batched NMS 0.7, top-1000,
RoI level assignment, score > 0.05, NMS 0.5, top-100.  A float32 evaluation that is as accurate but not bit-identical (the
library's default convolution numerics, or the reference's own cuDNN kernels) can legitimately flip a decision that sits on
its threshold.  This module evaluates the same path in THREE-VALUED logic: every decision whose margin is below `eps` is
"uncertain" and uncertainty propagates (a box that only an uncertain box would suppress is uncertain, too).  It returns

    certain   detections every evaluation within eps of this one must produce        (state K through the whole chain)
    possible  a superset: detections some evaluation within eps may produce            (K or U)

so that a parity test can assert   certain  <=  detections of the device  <=  possible   instead of "90 % re-found".
"""
from __future__ import annotations

import numpy as np

from . import detector as odet

f32 = np.float32
S, U, K = 0, 1, 2          # certainly absent / uncertain / certainly present


def _iou_matrix(b):
    """mmcv's IoU in float32 (area = w * h, no +1), [n][n]"""
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(f32)
    w = np.maximum(np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]), f32(0))
    h = np.maximum(np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]), f32(0))
    inter = (w * h).astype(f32)
    union = ((area[:, None] + area[None, :]).astype(f32) - inter).astype(f32)
    return np.where(union > 0, inter / np.maximum(union, f32(1e-30)), f32(0)).astype(f32)


def nms3(boxes, scores, cand, thr, eps_iou, eps_score):
    """Greedy NMS in three-valued logic.  cand[i] in {U, K}: is box i certainly a candidate.  Returns state[i] in {S, U, K}.
    Box j is certainly suppressed when a certainly-kept box that certainly precedes it (score higher by > eps_score) overlaps
    it by more than thr + eps_iou; possibly suppressed when any not-certainly-absent box that may precede it (score >
    s_j - eps_score, so near-ties count in both directions) overlaps it by more than thr - eps_iou."""
    n = len(boxes)
    state = np.full(n, S, np.int64)
    if n == 0:
        return state
    order = np.argsort(-scores, kind="stable")
    b, s, c = boxes[order].astype(f32), scores[order].astype(np.float64), np.asarray(cand)[order]
    iou = _iou_matrix(b).astype(np.float64)
    st = np.full(n, S, np.int64)
    for j in range(n):
        before = np.arange(n) < j
        surely_before = before & (s > s[j] + eps_score)
        maybe_before = (s >= s[j] - eps_score) & (np.arange(n) != j)         # includes (near-)ties that come later in this order
        later_tie = maybe_before & ~before
        if (surely_before & (st == K) & (iou[:, j] > thr + eps_iou)).any():
            st[j] = S
            continue
        poss = (before & maybe_before & (st != S) & (iou[:, j] > thr - eps_iou)).any() or \
               (later_tie & (iou[:, j] > thr - eps_iou)).any()
        st[j] = U if (poss or c[j] == U) else K
    state[order] = st
    return state


def _topk3(scores, state, k, eps_score):
    """keep the k best by score among the present boxes, three-valued: rank bounds from the certain / possible predecessors"""
    out = state.copy()
    idx = np.flatnonzero(state != S)
    if len(idx) == 0:
        return out
    s = scores[idx].astype(np.float64)
    for a, i in enumerate(idx):
        ahead_sure = int(((s > s[a] + eps_score) & (state[idx] == K)).sum())              # certainly present and certainly ahead
        ahead_poss = int(((s >= s[a] - eps_score) & (np.arange(len(idx)) != a)).sum())     # possibly present and possibly ahead
        if ahead_sure >= k:
            out[i] = S
        elif ahead_poss >= k:
            out[i] = U
    return out


def rpn_proposals3(cls_maps, reg_maps, eps_score=1e-4, eps_iou=1e-4, nms_pre=1000, max_per_img=1000, iou_thr=0.7):
    """oracle.detector.rpn_proposals in three-valued logic -> (boxes [P][4], scores [P], state [P] in {U, K}) over every
    proposal that is not certainly absent"""
    boxes_l, scores_l, ids_l, cand_l = [], [], [], []
    for lvl, (c, r) in enumerate(zip(cls_maps, reg_maps)):
        h, w, _ = c.shape
        scores = odet.sigmoid_f32(c.reshape(-1))
        deltas = r.reshape(-1, 4)
        anchors = odet.grid_anchors(h, w, odet.STRIDES[lvl])
        cand = np.full(scores.shape[0], K, np.int64)
        if 0 < nms_pre < scores.shape[0]:
            order = np.argsort(-scores, kind="stable")
            cut_in, cut_out = float(scores[order[nms_pre - 1]]), float(scores[order[nms_pre]])
            sel = np.flatnonzero(scores.astype(np.float64) >= cut_in - eps_score)           # possible members
            cand = np.where(scores[sel].astype(np.float64) > cut_out + eps_score, K, U)
            scores, deltas, anchors = scores[sel], deltas[sel], anchors[sel]
        boxes_l.append(odet.delta2bbox(anchors, deltas))
        scores_l.append(scores)
        ids_l.append(np.full(scores.shape[0], lvl, np.int64))
        cand_l.append(cand)
    boxes, scores, ids, cand = np.concatenate(boxes_l), np.concatenate(scores_l), np.concatenate(ids_l), np.concatenate(cand_l)
    valid = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    boxes, scores, ids, cand = boxes[valid], scores[valid], ids[valid], cand[valid]
    if len(boxes) == 0:
        return boxes, scores, cand
    off = (ids.astype(f32) * f32(boxes.max() + f32(1))).astype(f32)
    state = nms3((boxes + off[:, None]).astype(f32), scores, cand, iou_thr, eps_iou, eps_score)
    state = _topk3(scores, state, max_per_img, eps_score)
    keep = np.flatnonzero(state != S)
    keep = keep[np.argsort(-scores[keep], kind="stable")]
    return boxes[keep], scores[keep], state[keep]


def roi_levels3(rois, eps_rel=1e-5, num_levels=4, finest_scale=56):
    """SingleRoIExtractor.map_roi_levels with the set of levels a roi may be assigned to: [(level, certain)]"""
    scale = np.sqrt(((rois[:, 2] - rois[:, 0]).astype(f32) * (rois[:, 3] - rois[:, 1]).astype(f32)).astype(f32)).astype(np.float64)
    out = []
    for sc in scale:
        v = np.log2(sc / finest_scale + 1e-6)
        lv = {int(np.clip(np.floor(v + d), 0, num_levels - 1)) for d in (-eps_rel, 0.0, eps_rel)}
        out.append(sorted(lv))
    return out


def detect3(model, frame_wrapper_rgb, eps_score=1e-4, eps_iou=1e-4, score_thr=0.05, iou_thr=0.5, max_per_img=100):
    """oracle.detector.detect in three-valued logic.  Returns (certain [n][5], possible [m][5]); certain is a subset of possible
    and both are in descending score order."""
    x, sf, _ = odet.preprocess(frame_wrapper_rgb)
    feats = model.fpn(model.backbone(x[None]))
    cls_maps, reg_maps = model.rpn_head(feats)
    props, pscores, pstate = rpn_proposals3([c[0] for c in cls_maps], [r[0] for r in reg_maps], eps_score, eps_iou)
    if len(props) == 0:
        return np.zeros((0, 5), f32), np.zeros((0, 5), f32)
    levels = roi_levels3(props)
    rows_roi, rows_lv, rows_state = [], [], []
    for i, lvs in enumerate(levels):
        for lv in lvs:
            rows_roi.append(i)
            rows_lv.append(lv)
            rows_state.append(pstate[i] if len(lvs) == 1 else U)
    rows_roi, rows_lv, rows_state = np.array(rows_roi), np.array(rows_lv), np.array(rows_state)
    roi_feats = np.zeros((len(rows_roi), 7, 7, feats[0].shape[-1]), f32)
    for k, (i, lv) in enumerate(zip(rows_roi, rows_lv)):
        roi_feats[k] = odet.roi_align(feats[lv][0], props[i], 1.0 / odet.STRIDES[lv])
    cls, reg = model.roi_head(roi_feats)
    scores = odet.softmax_fg(cls)
    boxes = odet.delta2bbox(props[rows_roi], reg, stds=(0.1, 0.1, 0.2, 0.2))
    boxes = (boxes / sf[None, :].astype(f32)).astype(f32)
    sc64 = scores.astype(np.float64)
    cand = np.where(sc64 > score_thr + eps_score, rows_state, np.where(sc64 > score_thr - eps_score, U, S))
    sel = np.flatnonzero(cand != S)
    boxes, scores, cand = boxes[sel], scores[sel], cand[sel]
    if len(boxes) == 0:
        return np.zeros((0, 5), f32), np.zeros((0, 5), f32)
    state = nms3(boxes, scores, cand, iou_thr, eps_iou, eps_score)
    state = _topk3(scores, state, max_per_img, eps_score)
    order = np.argsort(-scores, kind="stable")
    rows = np.concatenate([boxes, scores[:, None]], axis=1).astype(f32)[order]
    st = state[order]
    return rows[st == K], rows[st != S]


def check_between(dets, certain, possible, box_tol, score_tol=1e-4):
    """certain <= dets <= possible, matching rows by box (max abs coordinate difference <= box_tol) and score.  Returns
    (missing certain rows, unexplained device rows, max box deviation of the matched certain rows)."""
    def match(row, pool):
        if len(pool) == 0:
            return -1, np.inf
        d = np.abs(pool[:, :4] - row[:4]).max(axis=1)
        d = np.where(np.abs(pool[:, 4] - row[4]) <= score_tol, d, np.inf)
        j = int(np.argmin(d))
        return (j, float(d[j])) if d[j] <= box_tol else (-1, float(d[j]))
    missing, worst = [], 0.0
    for row in certain:
        j, d = match(row, dets)
        if j < 0:
            missing.append(row)
        else:
            worst = max(worst, d)
    unexplained = [row for row in dets if match(row, possible)[0] < 0]
    return missing, unexplained, worst
