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
        return ids[:k].copy(), out[:k].copy(), info[:k].copy()

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
