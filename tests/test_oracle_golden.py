"""Pin the CPU oracle AND the host-side product logic against golden vectors produced by importing the
reference's own code (tests/golden/make_goldens.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

from oracle import boxes as obox
from oracle import decode as odec
from posepipeline_amd import keypoint_matching as km
from posepipeline_amd import tracking

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ---- Hungarian ------------------------------------------------------------------------------------
def test_linear_sum_assignment_matches_scipy_fixture():
    g = load("hungarian.npz")
    for k in range(int(g["n_cases"])):
        rows, cols = tracking.linear_sum_assignment(g[f"h{k}_cost"])
        assert np.array_equal(rows, g[f"h{k}_rows"]) and np.array_equal(cols, g[f"h{k}_cols"]), f"case {k}"


# ---- Kalman ---------------------------------------------------------------------------------------
def test_kalman_filter_trace():
    import ctypes as C
    from posepipeline_amd import _lib as L
    lib = L.load_library()
    g = load("deepsort.npz")
    meas, means, covs, gates = g["kf_meas"], g["kf_means"], g["kf_covs"], g["kf_gates"]
    mean = np.zeros(8)
    cov = np.zeros(64)
    lib.pp_kalman_initiate(L.ptr(np.ascontiguousarray(meas[0])), L.ptr(mean), L.ptr(cov))
    k = 0
    np.testing.assert_allclose(mean, means[k], rtol=1e-12)
    np.testing.assert_allclose(cov.reshape(8, 8), covs[k], rtol=1e-12)
    for i, z in enumerate(meas[1:]):
        lib.pp_kalman_predict(L.ptr(mean), L.ptr(cov))
        k += 1
        np.testing.assert_allclose(mean, means[k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(cov.reshape(8, 8), covs[k], rtol=1e-9, atol=1e-12)
        cand = np.ascontiguousarray(gates[i][:20].reshape(5, 4))
        out = np.zeros(5)
        lib.pp_kalman_gating_distance(L.ptr(mean), L.ptr(cov), L.ptr(cand), 5, L.ptr(out))
        np.testing.assert_allclose(out, gates[i][20:25], rtol=1e-9)
        lib.pp_kalman_update(L.ptr(mean), L.ptr(cov), L.ptr(np.ascontiguousarray(z)))
        k += 1
        np.testing.assert_allclose(mean, means[k], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cov.reshape(8, 8), covs[k], rtol=1e-8, atol=1e-12)


# ---- DeepSORT tracker traces ------------------------------------------------------------------------
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_deepsort_tracker_trace(case):
    g = load("deepsort.npz")
    p = f"c{case}_"
    det_off, trk_off = g[p + "det_off"], g[p + "trk_off"]
    trk = tracking.Tracker(mode=0, feat_dim=16)
    n_ids = set()
    for f in range(len(det_off) - 1):
        a, b = det_off[f], det_off[f + 1]
        ids, tlwh, info = trk.step(g[p + "det_tlwh"][a:b], g[p + "det_conf"][a:b], g[p + "det_feat"][a:b])
        ref = g[p + "trk_rows"][trk_off[f]:trk_off[f + 1]]
        assert len(ids) == len(ref), f"frame {f}: {len(ids)} tracks vs {len(ref)}"
        assert np.array_equal(ids, ref[:, 0].astype(np.int64)), f"frame {f}: ids {ids} vs {ref[:, 0]}"   # bit-exact ids
        assert np.array_equal(info, ref[:, 1:5].astype(np.int32)), f"frame {f}: state/hits/age/tsu"
        np.testing.assert_allclose(tlwh, ref[:, 5:9], rtol=1e-8, atol=1e-8)
        n_ids.update(ids.tolist())
    _, _, mean, cov = trk.dump()
    np.testing.assert_allclose(mean, ref[:, 9:17], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cov.reshape(len(ref), 64), ref[:, 17:], rtol=1e-6, atol=1e-9)
    assert len(n_ids) >= 1


# ---- NMS (deep_sort convention) -----------------------------------------------------------------------
def test_nms_deepsort_oracle():
    g = load("nms_deepsort.npz")
    for k in range(int(g["n_cases"])):
        scores = g[f"n{k}_scores"] if bool(g[f"n{k}_use_scores"]) else None
        pick = obox.nms_deepsort(g[f"n{k}_boxes"], float(g[f"n{k}_thr"]), scores)
        assert pick == g[f"n{k}_pick"].tolist(), f"case {k}"


# ---- PersonBbox.make ------------------------------------------------------------------------------------
def test_person_bbox_selection_and_smoothing():
    g = load("person_bbox.npz")
    for k in range(int(g["n_cases"])):
        off, rows = g[f"p{k}_off"], g[f"p{k}_rows"]
        tracks = [[{"track_id": int(r[0]), "tlhw": r[1:5]} for r in rows[off[i]:off[i + 1]]] for i in range(len(off) - 1)]
        bbox, present = tracking.person_bbox(tracks, g[f"p{k}_keep"])
        assert np.array_equal(present, g[f"p{k}_present"]), f"case {k}"
        assert np.array_equal(np.isnan(bbox), np.isnan(g[f"p{k}_bbox"]))
        assert np.array_equal(np.nan_to_num(bbox), np.nan_to_num(g[f"p{k}_bbox"])), f"case {k}"     # bit-exact selection


# ---- IoU / person match -----------------------------------------------------------------------------------
@pytest.mark.parametrize("mod", [obox, km])
def test_keypoint_matching(mod):
    g = load("keypoint_matching.npz")
    assert np.array_equal(mod.compute_iou(g["b1"], g["b2"]), g["iou_tlhw"])
    assert np.array_equal(mod.compute_iou(g["b1"], np.abs(g["b2"]), tlhw=False), g["iou_tlbr"])
    kps = list(g["kps"])
    assert np.array_equal(np.array([mod.keypoints_to_bbox(k) for k in kps], float), g["kp_bbox"])
    for bb, kp, idx in zip(g["match_bbox"], g["match_kp"], g["match_idx"]):
        got_kp, got_idx = mod.match_keypoints_to_bbox(bb, kps)
        assert (-1 if got_idx is None else got_idx) == idx
        assert np.array_equal(got_kp, kp)


# ---- DARK decode pieces ---------------------------------------------------------------------------------------
def test_dark_decode_against_in_tree_inference():
    g = load("dark_decode.npz")
    hms = g["heatmaps"]
    preds, maxvals = odec.get_max_preds(hms.astype(np.float32))
    ref_p, ref_m = g["preds"], g["maxvals"]
    pos = ref_m[..., 0] > 0
    assert np.array_equal(preds[pos], ref_p[pos].astype(np.float32))
    assert np.all(preds[~pos] == -1)          # mmpose marks empty maps with -1; the in-tree copy zeroes them
    assert np.all(ref_p[~pos] == 0)
    np.testing.assert_allclose(maxvals, ref_m, rtol=1e-6, atol=1e-30)   # 1e-63 underflows in float32
    # Taylor refinement: the reference ran in float64; the oracle follows mmpose's float32 heatmaps
    for name_in, name_out in (("heatmaps", "taylor"), ("log_noisy", "taylor_noisy")):
        logs = np.log(np.maximum(hms, 1e-10))[0] if name_in == "heatmaps" else g[name_in]
        for j in range(logs.shape[0]):
            start = ref_p[0, j].astype(np.float32)
            got = odec.taylor(logs[j].astype(np.float32), start.copy())
            assert np.abs(got - g[name_out][j]).max() < 2e-3, (name_out, j, got, g[name_out][j])
            # tight pin: the reference's taylor on the SAME float32-rounded logs (taylor*_f32in).  What is left is the
            # oracle's mmpose-style float32 differences against the in-tree float64 ones: <= 1e-4 px, ten times inside
            # the 1e-3 px budget this step has to protect
            got32 = odec.taylor(logs[j].astype(np.float32), start.copy())
            assert np.abs(got32 - g[name_out + "_f32in"][j]).max() <= 1e-4, (name_out, j, got32, g[name_out + "_f32in"][j])
    # analytic Gaussians away from the border are recovered (SURVEY.md 4.2 known-answer)
    cen = g["centres"]
    assert np.abs(g["taylor"][0] - cen[0]).max() < 1e-6
    # back-map: in-tree variant maps through the TLWH bbox; mmpose's centre/scale form is the same affine map
    bbox = g["bbox"]
    h, w = hms.shape[2:]
    center = (bbox[:2] + bbox[2:] / 2).astype(np.float32)
    scale = (bbox[2:] / 200.0).astype(np.float32)
    got = odec.transform_preds(g["taylor"].astype(np.float32), center, scale, [w, h])
    np.testing.assert_allclose(got, g["transformed"], atol=1e-3)


# ---- bbox aspect fix + VideoPose3D input normalisation ---------------------------------------------------------
def test_bbox_aspect_and_screen_normalisation():
    g = load("bbox_misc.npz")
    for name in ("default", "crop", "tight"):
        dil, ratio = g["args_" + name]
        got = np.array([obox.fix_bb_aspect_ratio(b, dilate=dil, ratio=ratio) for b in g["boxes"]])
        assert np.array_equal(got, g["fixed_" + name])
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    assert np.array_equal(normalize_screen_coordinates(g["nsc_X"], 1920, 1080), g["nsc_wide"])
    assert np.array_equal(normalize_screen_coordinates(g["nsc_X"], 1080, 1920), g["nsc_tall"])
