"""CPU oracle for box arithmetic.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

  * nms_deepsort: greedy NMS of pose_pipeline/wrappers/deep_sort_yolov4/deep_sort/preprocessing.py:5-70
    (tlwh boxes, +1 pixel areas :47, overlap = intersection / area of the OTHER box :66, suppress
    overlap > thr, picks in descending score order).  Pinned by tests/golden/nms_deepsort.npz.
  * nms_mmcv: the detector-side convention (mmcv-full `nms` / torchvision: x1y1x2y2, area = w*h, IoU,
    suppress IoU > thr, stable descending score order) selected by
    3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:101-109.  mmcv is not vendored: restated
    from its published algorithm, PARITY UNPINNED.
  * compute_iou / keypoints_to_bbox / match_keypoints_to_bbox: pose_pipeline/utils/keypoint_matching.py:4-68,
    pinned by tests/golden/keypoint_matching.npz.
  * fix_bb_aspect_ratio: pose_pipeline/utils/bounding_box.py:7-29, pinned by tests/golden/bbox_misc.npz.
"""
from __future__ import annotations

import numpy as np


def nms_deepsort(boxes_tlwh, max_overlap, scores=None):
    boxes = np.asarray(boxes_tlwh, dtype=np.float64).reshape(-1, 4)
    if len(boxes) == 0:
        return []
    x1, y1 = boxes[:, 0], boxes[:, 1]
    x2, y2 = boxes[:, 2] + boxes[:, 0], boxes[:, 3] + boxes[:, 1]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    idxs = list(np.argsort(scores if scores is not None else y2))
    pick = []
    while idxs:
        i = idxs.pop()
        pick.append(int(i))
        keep = []
        for j in idxs:
            w = max(0.0, min(x2[i], x2[j]) - max(x1[i], x1[j]) + 1)
            h = max(0.0, min(y2[i], y2[j]) - max(y1[i], y1[j]) + 1)
            if (w * h) / area[j] <= max_overlap:
                keep.append(j)
        idxs = keep
    return pick


def nms_mmcv(boxes_xyxy, scores, iou_thr):
    """float32 like the CUDA op; returns kept indices in descending score order."""
    b = np.asarray(boxes_xyxy, np.float32).reshape(-1, 4)
    s = np.asarray(scores, np.float32).reshape(-1)
    order = np.argsort(-s, kind="stable")
    b = b[order]
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    thr = np.float32(iou_thr)
    zero = np.float32(0)
    suppressed = np.zeros(len(b), bool)
    keep = []
    for i in range(len(b)):
        if suppressed[i]:
            continue
        keep.append(int(order[i]))
        r = slice(i + 1, len(b))
        w = np.maximum(np.minimum(b[i, 2], b[r, 2]) - np.maximum(b[i, 0], b[r, 0]), zero)
        h = np.maximum(np.minimum(b[i, 3], b[r, 3]) - np.maximum(b[i, 1], b[r, 1]), zero)
        inter = (w * h).astype(np.float32)
        # mmcv's devIoU compares without dividing: interS > threshold * (Sa + Sb - interS)
        suppressed[r] |= inter > (thr * ((area[i] + area[r]).astype(np.float32) - inter).astype(np.float32)).astype(np.float32)
    return keep


def compute_iou(box1, box2, tlhw=True, epsilon=1e-8):
    box1, box2 = np.asarray(box1, float), np.asarray(box2, float)
    n = max(box1.shape[0], box2.shape[0])
    b1p1, b1p2, b2p1, b2p2 = box1[:, :2], box1[:, 2:], box2[:, :2], box2[:, 2:]
    if tlhw:
        b1p2 = b1p1 + b1p2
        b2p2 = b2p1 + b2p2
    mask = np.ones(n) * np.all(b1p2 - b2p1 > 0, axis=1) * np.all(b2p2 - b1p1 > 0, axis=1)
    inter = np.prod(np.minimum(b2p2, b1p2) - np.maximum(b1p1, b2p1), axis=1)
    union = np.prod(b1p2 - b1p1, axis=1) + np.prod(b2p2 - b2p1, axis=1) - inter + epsilon
    return mask * (inter / union)


def keypoints_to_bbox(keypoints, thresh=0.1, min_keypoints=5):
    keypoints = np.asarray(keypoints)
    if keypoints.shape[-1] == 3:
        keypoints = keypoints[keypoints[:, -1] > thresh, :-1]
    if keypoints.shape[0] < min_keypoints:
        return [0.0, 0.0, 0.0, 0.0]
    x0, y0, x1, y1 = keypoints[:, 0].min(), keypoints[:, 1].min(), keypoints[:, 0].max(), keypoints[:, 1].max()
    return [x0, y0, x1 - x0, y1 - y0]


def match_keypoints_to_bbox(bbox, keypoints_list, thresh=0.25, num_keypoints=25, visible=True):
    empty = np.zeros((num_keypoints, 3 if visible else 2))
    if keypoints_list is None or len(keypoints_list) == 0:
        return empty, None
    kp_bbox = np.array([keypoints_to_bbox(k) for k in keypoints_list])
    iou = compute_iou(np.reshape(bbox, (1, 4)), kp_bbox)
    idx = int(np.argmax(iou))
    if iou[idx] > thresh:
        return keypoints_list[idx], idx
    return empty, None


def fix_bb_aspect_ratio(bbox, dilate=1.2, ratio=1.0):
    bbox = np.asarray(bbox, float)
    center = bbox[:2] + bbox[2:] / 2.0
    hw = bbox[2:]
    if hw[0] / hw[1] < ratio:
        hw = np.array([hw[1] * ratio, hw[1]])
    else:
        hw = np.array([hw[0], hw[0] / ratio])
    hw = hw * dilate
    return np.concatenate([center - hw / 2, hw], axis=0)
