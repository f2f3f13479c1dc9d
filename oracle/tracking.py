"""CPU oracle for the association stage.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

SortTrackerRef restates mmtrack 0.x `SortTracker.track` as configured by
3rdparty/mmtracking/mot/deepsort/sort_faster-rcnn_fpn_4e_mot17-private-half.py:16-18
(obj_score_thr 0.5, match_iou_thr 0.5, reid=None) and reached from pose_pipeline/wrappers/mmtrack.py:45:
drop detections with score <= 0.5; tracks seen in the previous frame are matched to the remaining
detections with scipy's Hungarian on 1 - IoU (mmdet `bbox_overlaps`, float32, eps 1e-6) and accepted when
the cost is < 0.5; everything else starts a new id from a running counter that begins at 0.  Without a
ReID model only last-frame tracks are candidates, so the Kalman state never influences the ids.
mmtrack is not vendored (requirements.txt:11 is a comment): PARITY UNPINNED against the real package.
The in-tree DeepSORT (mode 0 of the product tracker) needs no oracle of its own: the product is pinned
directly against traces of the imported reference (tests/golden/deepsort.npz).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

f32 = np.float32


def bbox_overlaps(a, b, eps=1e-6):
    a, b = a.astype(f32), b.astype(f32)
    area1 = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).astype(f32)
    area2 = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(f32)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.maximum((rb - lt).astype(f32), f32(0))
    overlap = (wh[..., 0] * wh[..., 1]).astype(f32)
    union = np.maximum((area1[:, None] + area2[None, :]).astype(f32) - overlap, f32(eps)).astype(f32)
    return (overlap / union).astype(f32)


class SortTrackerRef:
    def __init__(self, obj_score_thr=0.5, match_iou_thr=0.5):
        self.obj_score_thr, self.match_iou_thr = obj_score_thr, match_iou_thr
        self.num_tracks = 0
        self.last = []            # (id, box) of the tracks updated in the previous frame

    def step(self, dets):
        """dets [n][5] float32 (x1, y1, x2, y2, score) -> rows [m][6] float32 (id, x1, y1, x2, y2, score)"""
        dets = np.asarray(dets, f32).reshape(-1, 5)
        dets = dets[dets[:, 4] > f32(self.obj_score_thr)]
        ids = np.full(len(dets), -1, np.int64)
        if self.last and len(dets):
            tb = np.array([b for _, b in self.last], f32)
            dists = (f32(1) - bbox_overlaps(tb, dets[:, :4])).astype(f32)
            row, col = linear_sum_assignment(dists)
            for r, c in zip(row, col):
                if dists[r, c] < 1 - self.match_iou_thr:
                    ids[c] = self.last[r][0]
        for i in range(len(ids)):
            if ids[i] < 0:
                ids[i] = self.num_tracks
                self.num_tracks += 1
        self.last = [(int(i), d[:4].copy()) for i, d in zip(ids, dets)]
        return np.concatenate([ids[:, None].astype(f32), dets], axis=1).astype(f32)
