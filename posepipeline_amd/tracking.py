"""Host-side tracking stage: thin wrappers over pp_tracker_* plus the PersonBbox selection/smoothing.

Mirrors, for the cascade's tracking stage,
  * pose_pipeline/wrappers/deep_sort_yolov4/deep_sort/tracker.py (the in-tree DeepSORT, mode 0) and
    mmtrack's SortTracker as wired by wrappers/mmtrack.py:45 (mode 1) -- both in C++ behind the C ABI;
  * pose_pipeline/pipeline.py:656-687 `PersonBbox.make`: pick the frame's box iff exactly one of
    `keep_tracks` is present, then pandas `bfill(limit=2)` / `ffill(limit=2)` over missing rows.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def linear_sum_assignment(cost):
    """scipy.optimize.linear_sum_assignment restated in C++ (tie-breaking included)."""
    cost = np.ascontiguousarray(cost, np.float64)
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.zeros(max(n, 1), np.int32)
    cols = np.zeros(max(n, 1), np.int32)
    k = C.c_int32()
    L.check(L.load_library().pp_linear_sum_assignment(L.ptr(cost), nr, nc, L.ptr(rows), L.ptr(cols), C.byref(k)),
            "pp_linear_sum_assignment")
    return rows[: k.value].astype(np.int64), cols[: k.value].astype(np.int64)


class Tracker:
    """mode 0: in-tree DeepSORT (needs appearance features); mode 1: mmtrack SORT without ReID."""

    def __init__(self, mode=0, feat_dim=128, max_iou_distance=0.7, max_cosine_distance=0.3, max_age=30, n_init=3,
                 match_iou_thr=0.5, obj_score_thr=0.5):
        self.lib = L.load_library()
        self.mode = mode
        self.feat_dim = feat_dim if mode == 0 else 0
        h = C.c_void_p()
        a, b = (max_iou_distance, max_cosine_distance) if mode == 0 else (match_iou_thr, obj_score_thr)
        L.check(self.lib.pp_tracker_create(mode, self.feat_dim, a, b, max_age, n_init, C.byref(h)), "pp_tracker_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pp_tracker_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, tlwh, conf, feats=None, cap=1024):
        """-> (track_id int64 [n], tlwh float64 [n][4], info int32 [n][4])"""
        tlwh = np.ascontiguousarray(tlwh, np.float64).reshape(-1, 4)
        conf = np.ascontiguousarray(conf, np.float64).reshape(-1)
        n = tlwh.shape[0]
        f = None
        if self.mode == 0:
            f = np.ascontiguousarray(feats, np.float64).reshape(n, self.feat_dim)
        ids = np.zeros(cap, np.int64)
        out = np.zeros((cap, 4), np.float64)
        info = np.zeros((cap, 4), np.int32)
        k = C.c_int32()
        L.check(self.lib.pp_tracker_step(self.handle, L.ptr(tlwh), L.ptr(conf), L.ptr(f), n, cap, L.ptr(ids), L.ptr(out),
                                         L.ptr(info), C.byref(k)), "pp_tracker_step")
        k = k.value
        self._last_ids = set(int(i) for i in ids[:k])
        return ids[:k].copy(), out[:k].copy(), info[:k].copy()

    def live_ids(self) -> set:
        """ids that can still be reported by a later frame.  Both built modes report every track they keep: mode 0 emits
        all of tracker.tracks per frame (parser.py:76-86), mode 1 (no ReID) can only re-associate tracks seen in the
        previous frame -- so the live set is the id set of the last step."""
        return getattr(self, "_last_ids", set())

    def dump(self, cap=1024):
        ids = np.zeros(cap, np.int64)
        st = np.zeros((cap, 4), np.int32)
        mean = np.zeros((cap, 8))
        cov = np.zeros((cap, 64))
        k = C.c_int32()
        L.check(self.lib.pp_tracker_dump(self.handle, cap, L.ptr(ids), L.ptr(st), L.ptr(mean), L.ptr(cov), C.byref(k)),
                "pp_tracker_dump")
        k = k.value
        return ids[:k], st[:k], mean[:k], cov[:k].reshape(k, 8, 8)


def person_bbox(tracks, keep_tracks, limit=2):
    """PersonBbox.make (pipeline.py:656-687) without pandas.

    tracks: list (frames) of lists of dicts with "track_id" and "tlhw" (= x, y, w, h).
    Returns (bbox [N][4] float64 with NaN rows, present [N] bool)."""
    n = len(tracks)
    bbox = np.zeros((n, 4), np.float64)
    present = np.zeros(n, bool)
    keep = set(int(k) for k in np.atleast_1d(keep_tracks))
    for i, fr in enumerate(tracks):
        valid = [t for t in fr if int(t["track_id"]) in keep]
        if len(valid) == 1:
            present[i] = True
            bbox[i] = np.asarray(valid[0]["tlhw"], np.float64)
    out = bbox.copy()
    out[~present] = np.nan
    missing = ~present
    # bfill(limit): a NaN row takes the next valid row if it is among the `limit` NaNs directly before it
    filled = out.copy()
    nxt = -1
    run = 0
    fill_b = np.zeros(n, bool)
    for i in range(n - 1, -1, -1):
        if not missing[i]:
            nxt, run = i, 0
        else:
            run += 1
            if nxt >= 0 and run <= limit:
                filled[i] = out[nxt]
                fill_b[i] = True
    still = missing & ~fill_b
    res = filled.copy()
    prv = -1
    run = 0
    for i in range(n):
        if not still[i]:
            prv, run = i, 0
        else:
            run += 1
            if prv >= 0 and run <= limit:
                res[i] = filled[prv]
    return res, ~np.isnan(res).any(axis=1)


# ---- mmtrack ByteTracker (method "bytetrack" of wrappers/mmtrack.py) ------------------------------------------
def _iou_f32(a, b, eps=1e-6):
    """mmdet bbox_overlaps(mode='iou') in float32: a [n][4], b [m][4] -> [n][m]"""
    a, b = a.astype(np.float32), b.astype(np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = np.clip(np.minimum(a[:, None, 2:], b[None, :, 2:]) - np.maximum(a[:, None, :2], b[None, :, :2]), 0, None)
    overlap = wh[..., 0] * wh[..., 1]
    union = np.maximum(area_a[:, None] + area_b[None, :] - overlap, np.float32(eps))
    return overlap / union


class ByteTracker:
    """mmtrack ByteTracker as configured by 3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private-half.py:21-28
    (obj_score_thrs .6 / .1, init_track_thr .7, weight_iou_with_det_scores, match_iou_thrs .1 / .5 / .3, num_frames_retain 30).
    Kalman filter: pp_kalman_* (the C++ restatement pinned on the in-tree deep_sort filter, which mmtrack's KalmanFilter
    copies); assignment: `lap.lapjv(extend_cost=True, cost_limit)` = the optimum of the cost matrix extended with
    dummy rows / columns at cost_limit / 2, solved with pp_linear_sum_assignment.  mmtrack is not vendored: unpinned."""

    def __init__(self, high=0.6, low=0.1, init_track_thr=0.7, weight_iou_with_det_scores=True, match_iou_high=0.1,
                 match_iou_low=0.5, match_iou_tentative=0.3, num_frames_retain=30, num_tentatives=3):
        self.lib = L.load_library()
        self.high, self.low, self.init_thr = np.float32(high), np.float32(low), np.float32(init_track_thr)
        self.weight, self.thr = weight_iou_with_det_scores, (match_iou_high, match_iou_low, match_iou_tentative)
        self.retain, self.num_tentatives = num_frames_retain, num_tentatives
        self.tracks: dict = {}      # id -> [mean8, cov64, last_frame, hits, tentative]
        self.num_tracks = 0
        self.frame_id = -1

    def live_ids(self) -> set:
        return set(int(i) for i in self.tracks)

    # Kalman steps through the C ABI
    def _kf(self, fn, mean, cov, z=None):
        if z is None:
            L.check(getattr(self.lib, fn)(L.ptr(mean), L.ptr(cov)), fn)
        else:
            zz = np.ascontiguousarray(z, np.float64)
            L.check(getattr(self.lib, fn)(L.ptr(mean), L.ptr(cov), L.ptr(zz)), fn)

    @staticmethod
    def _cxcyah(box):
        b = box.astype(np.float32)
        w, h = b[2] - b[0], b[3] - b[1]
        return np.array([(b[2] + b[0]) / np.float32(2), (b[3] + b[1]) / np.float32(2), w / h, h], np.float64)

    def _track_boxes(self, ids):
        m = np.array([self.tracks[i][0][:4] for i in ids], np.float64).astype(np.float32).reshape(-1, 4)
        w = m[:, 2] * m[:, 3]
        two = np.float32(2)
        return np.stack([m[:, 0] - w / two, m[:, 1] - m[:, 3] / two, m[:, 0] + w / two, m[:, 1] + m[:, 3] / two], -1)

    def _assign(self, ids, dets, weight, thr):
        n, m = len(ids), len(dets)
        row, col = np.full(n, -1, np.int64), np.full(m, -1, np.int64)
        if n == 0 or m == 0:
            return row, col
        ious = _iou_f32(self._track_boxes(ids), dets[:, :4])
        if weight:
            ious = ious * dets[:, 4][None]
        dists = (np.float32(1) - ious).astype(np.float64)
        ext = np.full((n + m, n + m), (1 - thr) / 2.0)
        ext[n:, m:] = 0.0
        ext[:n, :m] = dists
        r, c = linear_sum_assignment(ext)
        for i, j in zip(r, c):
            if i < n and j < m:
                row[i], col[j] = j, i
        return row, col

    def step(self, dets):
        """dets [n][5] float32 (x1, y1, x2, y2, score) -> [m][6] float32 rows (id, x1, y1, x2, y2, score)"""
        self.frame_id += 1
        fid = self.frame_id
        dets = np.asarray(dets, np.float32).reshape(-1, 5)
        if not self.tracks or len(dets) == 0:
            out = dets[dets[:, 4] > self.init_thr]
            ids = np.arange(self.num_tracks, self.num_tracks + len(out), dtype=np.int64)
            self.num_tracks += len(out)
        else:
            first = dets[:, 4] > self.high
            second = (~first) & (dets[:, 4] > self.low)
            d1, d2 = dets[first], dets[second]
            confirmed = [i for i, t in self.tracks.items() if not t[4]]
            unconfirmed = [i for i, t in self.tracks.items() if t[4]]
            for i in confirmed:
                t = self.tracks[i]
                if t[2] != fid - 1:
                    t[0][7] = 0.0                       # lost in the previous frame: no vertical velocity
                self._kf("pp_kalman_predict", t[0], t[1])
            row1, col1 = self._assign(confirmed, d1, self.weight, self.thr[0])
            id1 = np.array([confirmed[r] if r > -1 else -1 for r in col1], np.int64)
            hit = id1 > -1
            u_b, u_i = d1[~hit], id1[~hit].copy()
            _, colt = self._assign(unconfirmed, u_b, self.weight, self.thr[2])
            for j, r in enumerate(colt):
                if r > -1:
                    u_i[j] = unconfirmed[r]
            rest = [i for k, i in enumerate(confirmed) if row1[k] == -1 and self.tracks[i][2] == fid - 1]
            _, col2 = self._assign(rest, d2, False, self.thr[1])
            id2 = np.array([rest[r] if r > -1 else -1 for r in col2], np.int64)
            keep2 = id2 > -1
            out = np.concatenate([d1[hit], u_b, d2[keep2]])
            ids = np.concatenate([id1[hit], u_i, id2[keep2]])
            new = ids == -1
            ids[new] = np.arange(self.num_tracks, self.num_tracks + int(new.sum()))
            self.num_tracks += int(new.sum())
        for i, b in zip(ids, out):
            i = int(i)
            z = self._cxcyah(b[:4])
            if i in self.tracks:
                t = self.tracks[i]
                self._kf("pp_kalman_update", t[0], t[1], z)
                t[2], t[3] = fid, t[3] + 1
                if t[4] and t[3] >= self.num_tentatives:
                    t[4] = False
            else:
                mean, cov = np.zeros(8), np.zeros(64)
                L.check(self.lib.pp_kalman_initiate(L.ptr(z), L.ptr(mean), L.ptr(cov)), "pp_kalman_initiate")
                self.tracks[i] = [mean, cov, fid, 1, fid != 0]
        for i in [i for i, t in self.tracks.items() if fid - t[2] >= self.retain or (t[4] and t[2] != fid)]:
            del self.tracks[i]
        if len(out) == 0:
            return np.zeros((0, 6), np.float32)
        return np.concatenate([ids[:, None].astype(np.float32), out], axis=1).astype(np.float32)
