#!/usr/bin/env python
"""Generate golden input/output vectors by IMPORTING the reference's own pure-numpy code.

Runs only in the build container, where /root/reference is mounted (the reference can never travel
to the GPU box); the .npz files it writes next to this script are committed and are what
tests/test_oracle_golden.py and the -m gpu parity tests read.  Nothing here copies reference source:
it calls the reference functions on seeded inputs and stores inputs + outputs.

Reference entry points exercised (paths relative to /root/reference/pose_pipeline):
  wrappers/deep_sort_yolov4/deep_sort/{tracker,track,kalman_filter,linear_assignment,iou_matching,
      nn_matching,preprocessing}.py   -- the in-tree DeepSORT tracker + greedy NMS
  pipeline.py:656-687                 -- PersonBbox.make (track selection + bfill/ffill smoothing)
  utils/keypoint_matching.py          -- keypoints_to_bbox, compute_iou, match_keypoints_to_bbox
  utils/inference.py                  -- get_max_preds, taylor, transform_preds (DARK, minus cv2 blur)
  utils/bounding_box.py:7-29          -- fix_bb_aspect_ratio
  wrappers/videopose3d.py:26-33       -- normalize_screen_coordinates (nested fn: formula re-evaluated)
plus scipy.optimize.linear_sum_assignment (the reference's Hungarian, linear_assignment.py:58).

usage: python tests/golden/make_goldens.py        (deterministic; rewrites the .npz files)
"""
import importlib
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ---- stubs for packages that are not installed here (cv2, datajoint) ---------------------------
def install_stubs():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "datajoint" in sys.modules:
        return
    dj = types.ModuleType("datajoint")
    dj.config = {"custom": {}}

    class _Rel:
        """Enough of a DataJoint relation for PersonBbox.make: `(Table & key).fetch1(attr)`."""
        rows = None

        def __and__(self, key):
            return self

        def fetch1(self, *attrs):
            vals = [type(self).rows[a] for a in attrs]
            return vals[0] if len(vals) == 1 else tuple(vals)

        def insert1(self, row, **kw):
            type(self).inserted = dict(row)

    class _Meta(type(_Rel)):
        def __and__(cls, key):
            return cls()

    class Table(_Rel, metaclass=_Meta):
        pass

    for name in ("Manual", "Lookup", "Computed", "Imported", "Part"):
        setattr(dj, name, type(name, (Table,), {}))

    def schema(*a, **k):
        def deco(cls):
            return cls
        return deco

    dj.schema = schema
    dj.Schema = schema
    sys.modules["datajoint"] = dj


def import_reference():
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pp = importlib.import_module("pose_pipeline")
    return pp


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- synthetic multi-person detection sequences --------------------------------------------------
def synth_sequence(rng, n_frames, n_people, feat_dim=16, drop=0.08, jitter=2.0, spurious=0.05):
    """Linear / crossing trajectories with births, deaths, gaps and false positives."""
    people = []
    for p in range(n_people):
        w = rng.uniform(60, 160)
        h = w * rng.uniform(2.0, 3.2)
        x0, y0 = rng.uniform(0, 1500), rng.uniform(0, 500)
        vx, vy = rng.uniform(-9, 9), rng.uniform(-2, 2)
        t0 = int(rng.integers(0, n_frames // 3))
        t1 = int(rng.integers(2 * n_frames // 3, n_frames + 1))
        f = rng.normal(size=feat_dim)
        people.append(dict(w=w, h=h, x0=x0, y0=y0, vx=vx, vy=vy, t0=t0, t1=t1, f=f / np.linalg.norm(f)))
    frames = []
    for t in range(n_frames):
        boxes, conf, feats = [], [], []
        for p in people:
            if not (p["t0"] <= t < p["t1"]) or rng.uniform() < drop:
                continue
            # a few long gaps (> max_age is exercised by t1/t0 re-entries below)
            bx = p["x0"] + p["vx"] * t + rng.uniform(-jitter, jitter)
            by = p["y0"] + p["vy"] * t + rng.uniform(-jitter, jitter)
            boxes.append([bx, by, p["w"] + rng.uniform(-jitter, jitter), p["h"] + rng.uniform(-jitter, jitter)])
            conf.append(rng.uniform(0.5, 1.0))
            f = p["f"] + rng.normal(scale=0.05, size=feat_dim)
            feats.append(f / np.linalg.norm(f))
        if rng.uniform() < spurious:
            boxes.append([rng.uniform(0, 1700), rng.uniform(0, 800), rng.uniform(40, 120), rng.uniform(80, 300)])
            conf.append(rng.uniform(0.5, 0.7))
            f = rng.normal(size=feat_dim)
            feats.append(f / np.linalg.norm(f))
        order = rng.permutation(len(boxes))
        frames.append((np.array(boxes, float).reshape(-1, 4)[order], np.array(conf, float)[order],
                       np.array(feats, float).reshape(-1, feat_dim)[order]))
    return frames


def gen_deepsort(pp):
    ds = "pose_pipeline.wrappers.deep_sort_yolov4.deep_sort"
    tracker_m = importlib.import_module(ds + ".tracker")
    nn = importlib.import_module(ds + ".nn_matching")
    det_m = importlib.import_module(ds + ".detection")
    kf_m = importlib.import_module(ds + ".kalman_filter")
    pre = importlib.import_module(ds + ".preprocessing")
    out = {}
    cases = [(0, 60, 1, 0.0), (1, 80, 3, 0.08), (2, 120, 6, 0.15), (3, 90, 4, 0.3)]
    out["n_cases"] = np.array(len(cases))
    for ci, (seed, n_frames, n_people, drop) in enumerate(cases):
        rng = np.random.default_rng(100 + seed)
        frames = synth_sequence(rng, n_frames, n_people, drop=drop)
        if ci == 3:  # a long occlusion (> max_age = 30 frames) -> track deletion and a fresh id
            for t in range(25, 62):
                frames[t] = (np.zeros((0, 4)), np.zeros((0,)), np.zeros((0, 16)))
        metric = nn.NearestNeighborDistanceMetric("cosine", 0.3, None)   # parser.py:35-47
        tracker = tracker_m.Tracker(metric)
        det_off, det_boxes, det_conf, det_feat = [0], [], [], []
        trk_off, trk_rows = [0], []
        for boxes, conf, feats in frames:
            dets = [det_m.Detection(b, c, 0, f) for b, c, f in zip(boxes, conf, feats)]
            # parser.py:66-70: NMS with nms_max_overlap = 1.0, then predict/update
            if len(dets):
                keep = pre.non_max_suppression(np.array([d.tlwh for d in dets]), 1.0, np.array([d.confidence for d in dets]))
                dets = [dets[i] for i in keep]
            det_boxes += [d.tlwh for d in dets]
            det_conf += [d.confidence for d in dets]
            det_feat += [d.feature for d in dets]
            det_off.append(len(det_boxes))
            tracker.predict()
            tracker.update(dets)
            for t in tracker.tracks:   # parser.py:76-86 emits every live track
                trk_rows.append([t.track_id, t.state, t.hits, t.age, t.time_since_update, *t.to_tlwh(), *t.mean,
                                 *t.covariance.reshape(-1)])
            trk_off.append(len(trk_rows))
        p = f"c{ci}_"
        out[p + "det_off"] = np.array(det_off)
        out[p + "det_tlwh"] = np.array(det_boxes, float).reshape(-1, 4)
        out[p + "det_conf"] = np.array(det_conf, float)
        out[p + "det_feat"] = np.array(det_feat, float).reshape(-1, 16)
        out[p + "trk_off"] = np.array(trk_off)
        out[p + "trk_rows"] = np.array(trk_rows, float).reshape(-1, 5 + 4 + 8 + 64)
    # Kalman filter on its own
    kf = kf_m.KalmanFilter()
    rng = np.random.default_rng(7)
    meas = np.array([[320.0, 240.0, 0.4, 200.0]]) + np.cumsum(rng.normal(scale=[3, 1, 0.002, 1], size=(12, 4)), 0)
    mean, cov = kf.initiate(meas[0])
    means, covs, gates = [mean], [cov], []
    for z in meas[1:]:
        mean, cov = kf.predict(mean, cov)
        means.append(mean), covs.append(cov)
        cand = z[None, :] + rng.normal(scale=[20, 20, 0.05, 15], size=(5, 4))
        gates.append(np.concatenate([cand.reshape(-1), kf.gating_distance(mean, cov, cand),
                                     kf.gating_distance(mean, cov, cand, only_position=True)]))
        mean, cov = kf.update(mean, cov, z)
        means.append(mean), covs.append(cov)
    out["kf_meas"] = meas
    out["kf_means"] = np.array(means)
    out["kf_covs"] = np.array(covs)
    out["kf_gates"] = np.array(gates)
    np.savez_compressed(os.path.join(OUT, "deepsort.npz"), **out)

    # greedy NMS (preprocessing.py:5-70)
    nms = {}
    rng = np.random.default_rng(3)
    k = 0
    for n in (0, 1, 7, 40, 150):
        base = rng.uniform(0, 400, (max(n // 4, 1), 2))
        boxes = np.concatenate([base[rng.integers(0, len(base), n)] + rng.uniform(-25, 25, (n, 2)),
                                rng.uniform(30, 120, (n, 2))], 1) if n else np.zeros((0, 4))
        scores = rng.uniform(0, 1, n)
        if n >= 7:
            scores[3] = scores[5]                      # a tie
            boxes[6] = boxes[2]                        # identical boxes
        for thr in (0.3, 0.5, 0.7, 1.0):
            for use_scores in (True, False):
                pick = pre.non_max_suppression(boxes.copy(), thr, scores if use_scores else None)
                nms[f"n{k}_boxes"], nms[f"n{k}_scores"] = boxes, scores
                nms[f"n{k}_thr"], nms[f"n{k}_use_scores"] = np.array(thr), np.array(use_scores)
                nms[f"n{k}_pick"] = np.array(pick, dtype=np.int64)
                k += 1
    nms["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "nms_deepsort.npz"), **nms)


def gen_hungarian():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(11)
    out, k = {}, 0
    for (r, c) in [(1, 1), (1, 5), (5, 1), (3, 3), (4, 7), (7, 4), (10, 10), (25, 40), (40, 25), (60, 60), (100, 100)]:
        for kind in ("uniform", "ties", "gated"):
            cost = rng.uniform(0, 1, (r, c))
            if kind == "ties":
                cost = np.round(cost * 4) / 4          # many equal costs -> tie-breaking order matters
            if kind == "gated":
                cost[cost > 0.7] = 0.7 + 1e-5          # linear_assignment.py:57
            rows, cols = linear_sum_assignment(cost)
            out[f"h{k}_cost"], out[f"h{k}_rows"], out[f"h{k}_cols"] = cost, rows, cols
            k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "hungarian.npz"), **out)


def gen_person_bbox(pp):
    rng = np.random.default_rng(21)
    out, k = {}, 0
    for n_frames, keep, pattern in [(40, [1], "gaps"), (60, [2, 5], "dups"), (30, [9], "never"), (25, [1], "edges"),
                                    (50, [3], "long_gap")]:
        tracks = []
        for t in range(n_frames):
            fr = []
            for tid in (1, 2, 3, 5):
                present = rng.uniform() > 0.25
                if pattern == "edges":
                    present = 3 <= t < n_frames - 4 and t not in (10, 11, 12)
                if pattern == "long_gap" and 20 <= t < 27:
                    present = False
                if present:
                    tlwh = np.array([100.0 * tid + 2 * t, 50.0 + t, 80.0, 200.0]) + rng.uniform(-1, 1, 4)
                    fr.append({"track_id": tid, "tlhw": tlwh, "tlbr": np.r_[tlwh[:2], tlwh[:2] + tlwh[2:]],
                               "confidence": float(rng.uniform(0.5, 1))})
            tracks.append(fr)
        pp.pipeline.TrackingBbox.rows = {"tracks": tracks}
        pp.pipeline.PersonBboxValid.rows = {"keep_tracks": keep}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pp.pipeline.PersonBbox().make({"video_subject_id": 0})
        res = pp.pipeline.PersonBbox.inserted
        # flatten the track list: per-frame offsets + rows [track_id, tlwh(4)]
        off, rows = [0], []
        for fr in tracks:
            rows += [[d["track_id"], *d["tlhw"]] for d in fr]
            off.append(len(rows))
        out[f"p{k}_off"], out[f"p{k}_rows"] = np.array(off), np.array(rows, float).reshape(-1, 5)
        out[f"p{k}_keep"] = np.array(keep)
        out[f"p{k}_bbox"], out[f"p{k}_present"] = np.asarray(res["bbox"], float), np.asarray(res["present"], bool)
        k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "person_bbox.npz"), **out)


def gen_keypoint_matching():
    km = load_by_path("ref_keypoint_matching", os.path.join(REF, "pose_pipeline/utils/keypoint_matching.py"))
    rng = np.random.default_rng(31)
    b1 = np.concatenate([rng.uniform(0, 300, (64, 2)), rng.uniform(0, 200, (64, 2))], 1)
    b2 = b1 + rng.uniform(-60, 60, (64, 4))
    b2[:, 2:] = np.abs(b2[:, 2:])
    b2[5] = b1[5]
    b2[6, 2:] = 0.0                                   # zero-size box
    b1[7, 2:] = 0.0
    out = {"b1": b1, "b2": b2, "iou_tlhw": km.compute_iou(b1, b2), "iou_tlbr": km.compute_iou(b1, np.abs(b2), tlhw=False)}
    kps = [np.concatenate([rng.uniform(0, 400, (25, 2)), rng.uniform(0, 1, (25, 1))], 1) for _ in range(6)]
    kps[2][:, 2] = 0.05                                # all below thresh -> zero bbox
    out["kps"] = np.array(kps)
    out["kp_bbox"] = np.array([km.keypoints_to_bbox(k) for k in kps], float)
    m_kp, m_idx = [], []
    for bb in [np.array(km.keypoints_to_bbox(kps[0])), np.array(km.keypoints_to_bbox(kps[4])) + 5,
               np.array([1000.0, 1000.0, 10.0, 10.0])]:
        kp, idx = km.match_keypoints_to_bbox(bb, kps)
        m_kp.append(kp), m_idx.append(-1 if idx is None else idx)
    out["match_bbox"] = np.array([np.array(km.keypoints_to_bbox(kps[0])), np.array(km.keypoints_to_bbox(kps[4])) + 5,
                                  np.array([1000.0, 1000.0, 10.0, 10.0])])
    out["match_kp"], out["match_idx"] = np.array(m_kp), np.array(m_idx)
    np.savez_compressed(os.path.join(OUT, "keypoint_matching.npz"), **out)


def gen_dark():
    inf = load_by_path("ref_inference", os.path.join(REF, "pose_pipeline/utils/inference.py"))
    rng = np.random.default_rng(41)
    h, w = 64, 48
    yy, xx = np.mgrid[0:h, 0:w].astype(float)
    centres = [(30.3, 40.7), (0.2, 0.6), (1.4, 50.0), (w - 2.3, 20.0), (w - 1.0, h - 1.0), (35.5, 1.2), (36.0, h - 2.6),
               (10.25, 10.75), (40.9, 60.1)]
    hms = np.stack([np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 3.0 ** 2)) for cx, cy in centres])
    hms = np.concatenate([hms, np.zeros((1, h, w)), np.full((1, h, w), 0.5), -hms[:1]])[None]   # zero, flat, negative
    preds, maxvals = inf.get_max_preds(hms.copy())
    logs = np.log(np.maximum(hms, 1e-10))
    tay = np.array([inf.taylor(logs[0, j], preds[0, j].copy()) for j in range(hms.shape[1])])
    noisy = logs[0] + rng.normal(scale=0.01, size=logs[0].shape)
    tay_noisy = np.array([inf.taylor(noisy[j], preds[0, j].copy()) for j in range(hms.shape[1])])
    bbox = np.array([120.5, 60.25, 211.0, 333.5])
    tp = inf.transform_preds(tay, bbox, [w, h])
    # the same Taylor step on float32-ROUNDED logs (held in float64 arrays): the inputs mmpose's float32 heat-maps give the
    # oracle, so that the comparison isolates the algorithm from the rounding of its input (tight pin: 1e-4 px)
    logs32 = logs[0].astype(np.float32).astype(np.float64)
    noisy32 = noisy.astype(np.float32).astype(np.float64)
    tay32 = np.array([inf.taylor(logs32[j], preds[0, j].copy()) for j in range(hms.shape[1])])
    tay_noisy32 = np.array([inf.taylor(noisy32[j], preds[0, j].copy()) for j in range(hms.shape[1])])
    np.savez_compressed(os.path.join(OUT, "dark_decode.npz"), heatmaps=hms, centres=np.array(centres), preds=preds,
                        maxvals=maxvals, taylor=tay, log_noisy=noisy, taylor_noisy=tay_noisy, bbox=bbox,
                        transformed=tp, taylor_f32in=tay32, taylor_noisy_f32in=tay_noisy32)


def gen_bbox_aspect(pp):
    # bounding_box.py imports `from pose_pipeline import Video, TrackingBbox, PersonBbox` -> needs the stubbed package
    bb = importlib.import_module("pose_pipeline.utils.bounding_box")
    rng = np.random.default_rng(51)
    boxes = np.concatenate([rng.uniform(0, 500, (32, 2)), rng.uniform(10, 300, (32, 2))], 1)
    out = {"boxes": boxes}
    for name, (dil, ratio) in {"default": (1.2, 1.0), "crop": (1.2, 288 / 384), "tight": (1.0, 0.75)}.items():
        out["fixed_" + name] = np.array([bb.fix_bb_aspect_ratio(b, dilate=dil, ratio=ratio) for b in boxes])
        out["args_" + name] = np.array([dil, ratio])
    # normalize_screen_coordinates is a nested function (wrappers/videopose3d.py:26-33): evaluate its formula
    X = rng.uniform(0, 1920, (10, 17, 2))
    out["nsc_X"] = X
    out["nsc_wide"] = X / 1920 * 2 - [1, 1080 / 1920]          # w > h branch
    out["nsc_tall"] = X / 1920 * 2 - [1080 / 1920, 1]          # else branch with (w, h) = (1080, 1920)
    np.savez_compressed(os.path.join(OUT, "bbox_misc.npz"), **out)


def main():
    pp = import_reference()
    gen_deepsort(pp)
    gen_hungarian()
    gen_person_bbox(pp)
    gen_keypoint_matching()
    gen_dark()
    gen_bbox_aspect(pp)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
