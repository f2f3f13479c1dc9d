"""CPU oracle for mmtrack's ByteTracker.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Configured by 3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private-half.py:21-28
(obj_score_thrs high .6 / low .1, init_track_thr .7, weight_iou_with_det_scores True, match_iou_thrs high .1 / low .5 /
tentative .3, num_frames_retain 30; motion = KalmanFilter) and reached from pose_pipeline/wrappers/mmtrack.py:45 with
method "bytetrack".  mmtrack (0.x) and its `lap` dependency are not vendored and not installed: PARITY UNPINNED.  The
restatement follows mmtrack's `ByteTracker.track`:
  first frame / empty tracker: detections with score > init_track_thr start tracks (confirmed at once on frame 0);
  otherwise: Kalman-predict every CONFIRMED track (vertical velocity zeroed when it was not seen in the previous
  frame); (1) confirmed tracks vs detections with score > high, cost 1 - IoU * score, limit 1 - 0.1; (2) tentative
  tracks vs the still unmatched high detections, limit 1 - 0.3; (3) confirmed tracks left unmatched that WERE seen in the
  previous frame vs detections with low < score <= high, cost 1 - IoU, limit 1 - 0.5; unmatched high detections start
  new (tentative) tracks, unmatched low ones are dropped.  Tentative tracks are confirmed after 3 hits and dropped on
  their first miss; confirmed tracks are dropped after num_frames_retain frames without a match.
`lap.lapjv(dists, extend_cost=True, cost_limit=L)` solves the assignment on the matrix extended with dummy rows /
columns of cost L/2, so a pair is matched iff it belongs to that optimum and costs < L: restated with scipy's solver on
the same extended matrix.  Kalman filter: the 8-state constant-velocity model of the in-tree
wrappers/deep_sort_yolov4/deep_sort/kalman_filter.py:14-217 (mmtrack's KalmanFilter is the same code).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

from .tracking import bbox_overlaps

f32 = np.float32


class KalmanRef:
    """kalman_filter.py:31-197 (x, y, a, h + velocities; std weights 1/20 and 1/160)"""

    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.wp, self.wv = 1.0 / 20, 1.0 / 160

    def initiate(self, z):
        mean = np.r_[z, np.zeros(4)]
        h = z[3]
        std = [2 * self.wp * h, 2 * self.wp * h, 1e-2, 2 * self.wp * h, 10 * self.wv * h, 10 * self.wv * h, 1e-5, 10 * self.wv * h]
        return mean, np.diag(np.square(std))

    def predict(self, mean, cov):
        h = mean[3]
        std = [self.wp * h, self.wp * h, 1e-2, self.wp * h, self.wv * h, self.wv * h, 1e-5, self.wv * h]
        q = np.diag(np.square(std))
        return self.F @ mean, np.linalg.multi_dot((self.F, cov, self.F.T)) + q

    def update(self, mean, cov, z):
        h = mean[3]
        std = [self.wp * h, self.wp * h, 1e-1, self.wp * h]
        pm = self.H @ mean
        pc = np.linalg.multi_dot((self.H, cov, self.H.T)) + np.diag(np.square(std))
        chol, lower = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
        gain = scipy.linalg.cho_solve((chol, lower), (cov @ self.H.T).T, check_finite=False).T
        return mean + (z - pm) @ gain.T, cov - np.linalg.multi_dot((gain, pc, gain.T))


def xyxy_to_cxcyah(b):
    """mmtrack bbox_xyxy_to_cxcyah on a float32 tensor (float32 arithmetic), widened for the float64 filter"""
    b = np.asarray(b, f32)
    w, h = f32(b[2] - b[0]), f32(b[3] - b[1])
    return np.array([f32(f32(b[2] + b[0]) / f32(2)), f32(f32(b[3] + b[1]) / f32(2)), f32(w / h), h], np.float64)


def cxcyah_to_xyxy(m):
    """the track mean converted to the detections' dtype (float32) first, then bbox_cxcyah_to_xyxy in float32"""
    cx, cy, a, h = (f32(v) for v in m[:4])
    w = f32(a * h)
    return np.array([cx - w / f32(2), cy - h / f32(2), cx + w / f32(2), cy + h / f32(2)], f32)


def lapjv_limited(dists, cost_limit):
    """lap.lapjv(dists, extend_cost=True, cost_limit=cost_limit) -> (row, col) with -1 for unmatched"""
    n, m = dists.shape
    row, col = np.full(n, -1, np.int64), np.full(m, -1, np.int64)
    if n == 0 or m == 0:
        return row, col
    ext = np.full((n + m, n + m), cost_limit / 2.0)
    ext[n:, m:] = 0.0
    ext[:n, :m] = dists
    r, c = linear_sum_assignment(ext)
    for i, j in zip(r, c):
        if i < n and j < m:
            row[i], col[j] = j, i
    return row, col


class ByteTrackerRef:
    def __init__(self, high=0.6, low=0.1, init_thr=0.7, weight_iou=True, thr_high=0.1, thr_low=0.5, thr_tentative=0.3,
                 num_frames_retain=30, num_tentatives=3):
        self.p = dict(high=high, low=low, init=init_thr, w=weight_iou, th=thr_high, tl=thr_low, tt=thr_tentative)
        self.retain, self.num_tentatives = num_frames_retain, num_tentatives
        self.kf = KalmanRef()
        self.tracks = {}          # id -> dict(mean, cov, last, hits, tentative)
        self.num_tracks = 0
        self.frame = -1

    def _assign(self, ids, dets, weight, thr):
        if not ids or len(dets) == 0:
            return np.full(len(ids), -1, np.int64), np.full(len(dets), -1, np.int64)
        tb = np.array([cxcyah_to_xyxy(self.tracks[i]["mean"]) for i in ids]).astype(f32)
        ious = bbox_overlaps(tb, dets[:, :4])
        if weight:
            ious = (ious * dets[:, 4][None].astype(f32)).astype(f32)
        dists = (f32(1) - ious).astype(f32)
        return lapjv_limited(dists.astype(np.float64), 1 - thr)

    def step(self, dets):
        """dets [n][5] float32 (x1, y1, x2, y2, score) in score order -> rows [m][6] (id, x1, y1, x2, y2, score)"""
        self.frame += 1
        fid = self.frame
        dets = np.asarray(dets, f32).reshape(-1, 5)
        p = self.p
        if not self.tracks or len(dets) == 0:
            out = dets[dets[:, 4] > f32(p["init"])]
            ids = np.arange(self.num_tracks, self.num_tracks + len(out))
            self.num_tracks += len(out)
        else:
            first = dets[:, 4] > f32(p["high"])
            second = (~first) & (dets[:, 4] > f32(p["low"]))
            d1, d2 = dets[first], dets[second]
            confirmed = [i for i, t in self.tracks.items() if not t["tentative"]]
            unconfirmed = [i for i, t in self.tracks.items() if t["tentative"]]
            for i in confirmed:
                t = self.tracks[i]
                if t["last"] != fid - 1:
                    t["mean"][7] = 0
                t["mean"], t["cov"] = self.kf.predict(t["mean"], t["cov"])
            id1 = np.full(len(d1), -1, np.int64)
            row1, col1 = self._assign(confirmed, d1, p["w"], p["th"])
            for j, r in enumerate(col1):
                if r > -1:
                    id1[j] = confirmed[r]
            matched = id1 > -1
            m_b, m_i = d1[matched], id1[matched]
            u_b, u_i = d1[~matched], id1[~matched].copy()
            _, colt = self._assign(unconfirmed, u_b, p["w"], p["tt"])
            for j, r in enumerate(colt):
                if r > -1:
                    u_i[j] = unconfirmed[r]
            rest = [i for k, i in enumerate(confirmed) if row1[k] == -1 and self.tracks[i]["last"] == fid - 1]
            id2 = np.full(len(d2), -1, np.int64)
            _, col2 = self._assign(rest, d2, False, p["tl"])
            for j, r in enumerate(col2):
                if r > -1:
                    id2[j] = rest[r]
            keep2 = id2 > -1
            out = np.concatenate([m_b, u_b, d2[keep2]])
            ids = np.concatenate([m_i, u_i, id2[keep2]])
            new = ids == -1
            ids[new] = np.arange(self.num_tracks, self.num_tracks + int(new.sum()))
            self.num_tracks += int(new.sum())
        for i, b in zip(ids, out):
            i = int(i)
            z = xyxy_to_cxcyah(b[:4])
            if i in self.tracks:
                t = self.tracks[i]
                t["mean"], t["cov"] = self.kf.update(t["mean"], t["cov"], z)
                t["last"], t["hits"] = fid, t["hits"] + 1
                if t["tentative"] and t["hits"] >= self.num_tentatives:
                    t["tentative"] = False
            else:
                mean, cov = self.kf.initiate(z)
                self.tracks[i] = dict(mean=mean, cov=cov, last=fid, hits=1, tentative=fid != 0)
        for i in [i for i, t in self.tracks.items()
                  if fid - t["last"] >= self.retain or (t["tentative"] and t["last"] != fid)]:
            del self.tracks[i]
        return np.concatenate([ids[:, None].astype(f32), out], axis=1).astype(f32) if len(out) else np.zeros((0, 6), f32)
