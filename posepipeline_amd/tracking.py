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


# ---- mmtrack SortTracker WITH its ReID branch (method "deepsort" of wrappers/mmtrack.py) ---------------------------------
REID_GATED = 1e6      # surrogate cost of a Kalman-gated (track, detection) pair in the appearance assignment, see below


def reid_cdist(a, b):
    """torch.cdist(a, b) (p = 2) of float32 rows, defined here as: float64 sum of squared float32 differences, sqrt,
    rounded to float32"""
    d = a.astype(np.float32)[:, None, :].astype(np.float64) - b.astype(np.float32)[None, :, :].astype(np.float64)
    return np.sqrt((d * d).sum(-1)).astype(np.float32)


class SortReidTracker:
    """mmtrack 0.x `SortTracker.track` as configured by 3rdparty/mmtracking/mot/deepsort/
    deepsort_faster-rcnn_fpn_4e_mot17-private-half.py:43-54 (obj_score_thr .5, reid num_samples 10 / match_score_thr 2.0,
    match_iou_thr .5, num_tentatives 2, num_frames_retain 100; motion = KalmanFilter(center_only=False)):

      frame with no track yet (or no kept detection): every kept detection starts a track (tentative until it has
      `num_tentatives` boxes; a tentative track that misses a frame is dropped), ids from a running counter starting at 0;
      otherwise: Kalman-predict EVERY track and gate it against every detection (chi-square 0.95, 4 dof);
        1. appearance: confirmed tracks (any age up to num_frames_retain) vs detections, Euclidean distance between the mean
           of the track's last <= 10 embeddings and the detection's embedding, gated pairs excluded, Hungarian, accept
           distance <= 2.0;
        2. IoU: tracks updated in the previous frame and not matched yet vs the unmatched detections, cost 1 - IoU of the
           last OBSERVED box (float32), Hungarian, accept cost < 0.5;
      unmatched detections start new tracks; matched tracks get a Kalman update with the detection.

    The gated pairs: mmtrack writes NaN into the distance matrix before scipy.optimize.linear_sum_assignment, which current
    scipy rejects ("matrix contains invalid numeric entries") and old scipy resolved by unspecified NaN comparisons.  The
    well-defined reading implemented here (and in oracle/reid_mm.py): a gated pair costs REID_GATED = 1e6 in the assignment
    -- so the optimum uses as few gated pairs as possible -- and is never accepted.
    Kalman filter / Hungarian: the C++ restatements behind pp_kalman_* / pp_linear_sum_assignment (fixture-pinned on the
    in-tree deep_sort filter, which mmtrack's KalmanFilter copies).  mmtrack is not vendored: PARITY UNPINNED."""

    CHI2INV95_4 = 9.4877

    def __init__(self, obj_score_thr=0.5, match_iou_thr=0.5, match_score_thr=2.0, num_samples=10, num_tentatives=2,
                 num_frames_retain=100):
        self.lib = L.load_library()
        self.obj_score_thr, self.match_iou_thr, self.match_score_thr = np.float32(obj_score_thr), match_iou_thr, match_score_thr
        self.num_samples, self.num_tentatives, self.retain = num_samples, num_tentatives, num_frames_retain
        self.tracks: dict = {}      # id -> dict(mean, cov, box, last, n, tentative, embeds); insertion-ordered like mmtrack's
        self.num_tracks = 0
        self.frame_id = -1

    def live_ids(self) -> set:
        return set(self.tracks)

    def keep(self, dets):
        """mask of the detections the tracker uses (score > obj_score_thr): only these need an embedding"""
        return np.asarray(dets, np.float32).reshape(-1, 5)[:, 4] > self.obj_score_thr

    def _kf(self, fn, *args):
        L.check(getattr(self.lib, fn)(*[L.ptr(a) for a in args]), fn)

    def step(self, dets, embeds):
        """dets [n][5] float32 (x1, y1, x2, y2, score), ALREADY filtered with keep(); embeds [n][d] float32
        -> rows [n][6] float32 (id, x1, y1, x2, y2, score)"""
        self.frame_id += 1
        fid = self.frame_id
        dets = np.asarray(dets, np.float32).reshape(-1, 5)
        n = len(dets)
        embeds = np.asarray(embeds, np.float32).reshape(n, -1) if n else np.zeros((0, 1), np.float32)
        zs = np.stack([ByteTracker._cxcyah(d[:4]) for d in dets]) if n else np.zeros((0, 4))
        ids = np.full(n, -1, np.int64)
        if self.tracks and n:
            order = list(self.tracks)
            gate = np.zeros((len(order), n))
            for r, i in enumerate(order):
                t = self.tracks[i]
                self._kf("pp_kalman_predict", t["mean"], t["cov"])
                out = np.zeros(n)
                L.check(self.lib.pp_kalman_gating_distance(L.ptr(t["mean"]), L.ptr(t["cov"]), L.ptr(np.ascontiguousarray(zs)), n, L.ptr(out)),
                        "pp_kalman_gating_distance")
                gate[r] = out
            gated = gate > self.CHI2INV95_4
            confirmed = [i for i in order if not self.tracks[i]["tentative"]]
            if confirmed:
                means = []
                for i in confirmed:
                    e = self.tracks[i]["embeds"][-self.num_samples:]
                    acc = np.zeros_like(e[0])
                    for v in e:
                        acc = (acc + v).astype(np.float32)
                    means.append((acc / np.float32(len(e))).astype(np.float32))
                dist = reid_cdist(np.stack(means), embeds)
                g = gated[[order.index(i) for i in confirmed]]
                cost = np.where(g, REID_GATED, dist.astype(np.float64))
                rows, cols = linear_sum_assignment(cost)
                for r, c in zip(rows, cols):
                    if not g[r, c] and dist[r, c] <= self.match_score_thr:
                        ids[c] = confirmed[r]
            active = [i for i in order if i not in set(ids.tolist()) and self.tracks[i]["last"] == fid - 1]
            if active:
                free = np.flatnonzero(ids == -1)
                if len(free):
                    tb = np.stack([self.tracks[i]["box"] for i in active])
                    dists = (np.float32(1) - _iou_f32(tb, dets[free, :4])).astype(np.float64)
                    rows, cols = linear_sum_assignment(dists)
                    for r, c in zip(rows, cols):
                        if dists[r, c] < 1 - self.match_iou_thr:
                            ids[free[c]] = active[r]
        new = ids == -1
        ids[new] = np.arange(self.num_tracks, self.num_tracks + int(new.sum()))
        self.num_tracks += int(new.sum())
        for i, d, z, e in zip(ids.tolist(), dets, zs, embeds):
            if i in self.tracks:
                t = self.tracks[i]
                self._kf("pp_kalman_update", t["mean"], t["cov"], np.ascontiguousarray(z))
                t["box"], t["last"], t["n"] = d[:4].copy(), fid, t["n"] + 1
                t["embeds"] = (t["embeds"] + [e.copy()])[-self.num_samples:]
                if t["tentative"] and t["n"] >= self.num_tentatives:
                    t["tentative"] = False
            else:
                mean, cov = np.zeros(8), np.zeros(64)
                L.check(self.lib.pp_kalman_initiate(L.ptr(np.ascontiguousarray(z)), L.ptr(mean), L.ptr(cov)), "pp_kalman_initiate")
                self.tracks[i] = dict(mean=mean, cov=cov, box=d[:4].copy(), last=fid, n=1, tentative=True, embeds=[e.copy()])
        for i in [i for i, t in self.tracks.items() if fid - t["last"] >= self.retain or (t["tentative"] and t["last"] != fid)]:
            del self.tracks[i]
        return np.concatenate([ids[:, None].astype(np.float32), dets], axis=1).astype(np.float32) if n else np.zeros((0, 6), np.float32)
