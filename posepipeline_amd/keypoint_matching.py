"""Host glue next to the hot path: IoU in the reference's `tlhw` convention and best-IoU person match.

Same names, argument meaning and return values as pose_pipeline/utils/keypoint_matching.py:4-68
(used by DetectedFrames / bottom-up matching around the cascade); vectorised float64 numpy.
"""
from __future__ import annotations

import numpy as np


def keypoints_to_bbox(keypoints, thresh=0.1, min_keypoints=5):
    keypoints = np.asarray(keypoints)
    if keypoints.shape[-1] == 3:
        keypoints = keypoints[keypoints[:, -1] > thresh, :-1]
    if keypoints.shape[0] < min_keypoints:
        return [0.0, 0.0, 0.0, 0.0]
    lo = keypoints.min(axis=0)
    hi = keypoints.max(axis=0)
    return [lo[0], lo[1], hi[0] - lo[0], hi[1] - lo[1]]


def compute_iou(box1: np.ndarray, box2: np.ndarray, tlhw=True, epsilon=1e-8):
    """IoU of paired boxes ((N,4) vs (N,4) or (1,4) broadcast); 0 where they do not overlap."""
    box1 = np.asarray(box1, dtype=float)
    box2 = np.asarray(box2, dtype=float)
    tl1, br1 = box1[:, :2], box1[:, 2:]
    tl2, br2 = box2[:, :2], box2[:, 2:]
    if tlhw:
        br1 = tl1 + br1
        br2 = tl2 + br2
    overlap = np.all(br1 - tl2 > 0, axis=1) & np.all(br2 - tl1 > 0, axis=1)
    inter = np.prod(np.minimum(br2, br1) - np.maximum(tl1, tl2), axis=1)
    union = np.prod(br1 - tl1, axis=1) + np.prod(br2 - tl2, axis=1) - inter + epsilon
    return overlap.astype(float) * (inter / union)


def match_keypoints_to_bbox(bbox: np.ndarray, keypoints_list: list, thresh=0.25, num_keypoints=25, visible=True):
    """Best-IoU keypoint set for `bbox`, or zeros + None when nothing clears `thresh`."""
    empty = np.zeros((num_keypoints, 3 if visible else 2))
    if keypoints_list is None or len(keypoints_list) == 0:
        return empty, None
    boxes = np.array([keypoints_to_bbox(k) for k in keypoints_list])
    iou = compute_iou(np.reshape(bbox, (1, 4)), boxes)
    idx = int(np.argmax(iou))
    if iou[idx] > thresh:
        return keypoints_list[idx], idx
    return empty, None
