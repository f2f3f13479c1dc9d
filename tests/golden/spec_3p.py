"""Shared by tests/golden/make_goldens_3p.py (the generator that calls the REAL third-party functions, wherever they are
installed) and tests/test_oracle_3p.py (which holds oracle/ to the fixtures when they are present).

Every [3P] stage of the path (cv2, mmcv, mmpose 0.x, mmdet 2.x, mmtrack 0.x, VideoPose3D) is one SECTION:
    inputs(rng)      seeded inputs (small: the fixtures are committed data, a few hundred KB in total)
    oracle(inputs)   the same computation through oracle/ (this repository's CPU restatement)
    checks           per output key: "exact" or an absolute tolerance
A fixture file tests/golden/3p_<section>.npz holds the inputs and the third-party outputs under the same keys.
Inputs are regenerated from the seed on both sides, and also stored, so a fixture is self-contained data.
"""
from __future__ import annotations

import numpy as np

SEED = 20240901


def _frames(rng, n, h, w):
    base = rng.integers(0, 256, (n, h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :h, :w]
    return np.clip(fr.astype(np.int64) + rng.integers(-20, 21, fr.shape), 0, 255).astype(np.uint8)


BOXES = np.array([                       # the cases of tests/test_gpu_stages.py::test_crop_affine_normalize_bit_exact, scaled
    [50.3, 20.7, 40.2, 95.9],            # inside
    [-15.5, -10.25, 100.0, 75.0],        # over the top-left corner: zero border taps
    [200.0, 100.0, 60.0, 50.0],          # over the bottom-right corner, wide box
    [5.0, 5.0, 15.0, 30.0],              # small box: magnification
    [0.0, 0.0, 240.0, 135.0],            # whole frame: minification
], np.float64)


# ---- cv2: getAffineTransform + warpAffine (mmpose TopDownAffine) -----------------------------------------------------
def cv2_affine_inputs(rng):
    from oracle import preprocess as opre
    frames = _frames(rng, 2, 135, 240)
    cases = [(i, i % 2, (72, 96)) for i in range(len(BOXES))] + [(0, 0, (288, 384))]
    d = {"frames": frames, "n_cases": np.int64(len(cases))}
    for k, (bi, fi, size) in enumerate(cases):
        c, s = opre.box2cs(BOXES[bi], size)
        src, dst = opre.affine_points(c, s, size)
        d[f"c{k}_frame"], d[f"c{k}_size"], d[f"c{k}_src"], d[f"c{k}_dst"] = np.int64(fi), np.array(size, np.int64), src, dst
    return d


def cv2_affine_oracle(d, ref=None):
    """ref: the fixture (its cv2 matrix is used for the warp, so that the two functions are pinned separately)"""
    from oracle import preprocess as opre
    out = {}
    for k in range(int(d["n_cases"])):
        m = opre.get_affine_transform_cv(d[f"c{k}_src"], d[f"c{k}_dst"])
        out[f"c{k}_trans"] = m
        m_warp = ref[f"c{k}_trans"] if ref is not None else m
        out[f"c{k}_crop"] = opre.warp_affine_u8(d["frames"][int(d[f"c{k}_frame"])], m_warp, tuple(int(v) for v in d[f"c{k}_size"]))
        out[f"c{k}_crop_own_matrix"] = opre.warp_affine_u8(d["frames"][int(d[f"c{k}_frame"])], m, tuple(int(v) for v in d[f"c{k}_size"]))
    return out


def cv2_affine_checks(d):
    ch = {}
    for k in range(int(d["n_cases"])):
        ch[f"c{k}_trans"] = 1e-9
        ch[f"c{k}_crop"] = "exact"
    return ch


# ---- cv2: GaussianBlur float32 (mmpose _gaussian_blur, kernels 17 and 11) ------------------------------------------------
def cv2_blur_inputs(rng):
    h, w = 40, 30
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.zeros((4, h, w), np.float32)
    for i, (cy, cx, s) in enumerate([(20.3, 14.6, 2.0), (1.2, 2.7, 1.5), (38.4, 28.9, 3.0), (10.0, 22.0, 1.0)]):
        hm[i] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)).astype(np.float32)
    hm += rng.uniform(0, 0.02, hm.shape).astype(np.float32)
    return {"hm": hm}


def cv2_blur_oracle(d, ref=None):
    from oracle import decode as odec
    out = {}
    for k in (17, 11):
        out[f"blur{k}"] = np.stack([odec.gaussian_blur_f32(m, k) for m in d["hm"]])
        out[f"kernel{k}"] = odec.gaussian_kernel1d(k)
    return out


def cv2_blur_checks(d):
    return {"blur17": "exact", "blur11": "exact", "kernel17": "exact", "kernel11": "exact"}


# ---- cv2.resize INTER_LINEAR 8-bit (mmcv.imrescale of the detector pipeline) ----------------------------------------------
def cv2_resize_inputs(rng):
    return {"img": _frames(rng, 1, 135, 240)[0], "dsize": np.array([136, 77], np.int64), "dsize_up": np.array([300, 169], np.int64)}


def cv2_resize_oracle(d, ref=None):
    from oracle import detector as odet
    return {"down": odet.resize_linear_u8(d["img"], tuple(int(v) for v in d["dsize"])),
            "up": odet.resize_linear_u8(d["img"], tuple(int(v) for v in d["dsize_up"]))}


def cv2_resize_checks(d):
    return {"down": "exact", "up": "exact"}


# ---- cv2.cvtColor(COLOR_YUV2BGR_NV12) 8-bit (the NV12 frame source: csrc/nv12.hip, oracle/nv12.py) ---------------------------
def cv2_nv12_inputs(rng):
    h, w = 36, 40
    planes = rng.integers(0, 256, (3, h * 3 // 2, w)).astype(np.uint8)       # every byte value, legal range or not
    planes[0, :2, :8] = [[0, 255, 16, 235, 81, 145, 41, 128], [15, 236, 128, 1, 90, 54, 240, 200]]
    return {"planes": planes, "hw": np.array([h, w], np.int64)}


def cv2_nv12_oracle(d, ref=None):
    from oracle import nv12 as onv
    return {"bgr": onv.nv12_to_bgr(d["planes"], int(d["hw"][0]), int(d["hw"][1]))}


def cv2_nv12_checks(d):
    return {"bgr": "exact"}


# ---- mmpose 0.x: _box2cs, get_affine_transform, flip_back, keypoints_from_heatmaps, transform_preds ------------------------
def mmpose_inputs(rng):
    n, k, h, w = 2, 17, 64, 48
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = rng.uniform(0, 0.01, (n, k, h, w)).astype(np.float32)
    for i in range(n):
        for j in range(k):
            cy, cx = rng.uniform(0, h), rng.uniform(0, w)          # includes peaks at / near the borders
            hm[i, j] += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 8.0).astype(np.float32)
    hm[1, 3] = 0.0                                                  # an all-zero map (maxval <= 0 -> preds -1)
    hmf = hm[:, :, :, ::-1].copy() + rng.uniform(0, 0.01, hm.shape).astype(np.float32)
    center = np.array([[120.5, 80.25], [300.0, 200.0]], np.float32)
    scale = np.array([[0.9, 1.2], [2.4, 3.2]], np.float32)
    return {"boxes": BOXES, "hm": hm, "hm_flipped": hmf, "center": center, "scale": scale}


def mmpose_oracle(d, ref=None):
    from oracle import decode as odec
    from oracle import preprocess as opre
    from posepipeline_amd.models import hrnet
    out = {}
    for i, bb in enumerate(d["boxes"]):
        c, s = opre.box2cs(bb, (288, 384))
        out[f"box2cs{i}"] = np.concatenate([c, s])
        src, dst = opre.affine_points(c, s, (288, 384))
        out[f"affine{i}"] = opre.get_affine_transform_cv(src, dst)
    out["merged"] = odec.flip_merge(d["hm"], d["hm_flipped"], hrnet.COCO_FLIP_PAIRS, shift_heatmap=True)
    for post in ("unbiased", "default"):
        p, m = odec.keypoints_from_heatmaps(d["hm"], d["center"], d["scale"], post_process=post, kernel=17 if post == "unbiased" else 11)
        out[f"preds_{post}"], out[f"maxvals_{post}"] = p, m
    out["transform_preds"] = odec.transform_preds(np.array([[0.0, 0.0], [47.0, 63.0], [10.25, 20.75]], np.float32), d["center"][0], d["scale"][0], [48, 64])
    return out


def mmpose_checks(d):
    ch = {f"box2cs{i}": "exact" for i in range(len(d["boxes"]))}
    ch.update({f"affine{i}": 1e-9 for i in range(len(d["boxes"]))})
    # the key-point bar of north_star is 1e-3 px; everything that is not a transcendental is held to exact equality
    ch.update({"merged": "exact", "maxvals_unbiased": "exact", "maxvals_default": "exact", "preds_default": "exact",
               "preds_unbiased": 1e-3, "transform_preds": "exact"})
    return ch


# ---- mmcv-full 1.x: nms, roi_align (aligned=True, sampling_ratio=0), imnormalize -----------------------------------------
def mmcv_inputs(rng):
    n = 300
    c = rng.uniform(0, 600, (40, 2))
    ctr = c[rng.integers(0, 40, n)] + rng.normal(0, 10, (n, 2))
    wh = rng.uniform(20, 160, (n, 2))
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    feat = rng.standard_normal((1, 8, 40, 68)).astype(np.float32)              # NCHW like mmcv
    rois = np.array([[0, 10.3, 20.7, 200.9, 180.2], [0, -20.0, -8.0, 90.0, 60.0], [0, 400.0, 250.0, 543.9, 319.9],
                     [0, 100.0, 100.0, 101.0, 101.5], [0, 0.0, 0.0, 544.0, 320.0]], np.float32)
    img = _frames(rng, 1, 64, 96)[0]
    return {"boxes": boxes, "scores": scores, "feat": feat, "rois": rois, "img": img}


def mmcv_oracle(d, ref=None):
    from oracle import boxes as obox
    from oracle import detector as odet
    out = {f"keep_{int(t * 10)}": np.array(obox.nms_mmcv(d["boxes"], d["scores"], t), np.int64) for t in (0.5, 0.7)}
    feat = np.ascontiguousarray(np.transpose(d["feat"][0], (1, 2, 0)))                # oracle works NHWC
    out["roi_align_s8"] = np.stack([np.transpose(odet.roi_align(feat, r[1:], 1.0 / 8.0), (2, 0, 1)) for r in d["rois"]])
    lut = odet.normalize_lut()
    out["imnormalize"] = np.stack([lut[c][d["img"][:, :, 2 - c]] for c in range(3)], axis=-1)     # to_rgb=True
    return out


def mmcv_checks(d):
    return {"keep_5": "exact", "keep_7": "exact", "roi_align_s8": 1e-5, "imnormalize": "exact"}


# ---- mmtrack 0.x: SortTracker.track without ReID (sort_faster-rcnn config) ---------------------------------------------------
def mmtrack_inputs(rng):
    n_frames = 24
    dets, counts = np.zeros((n_frames, 6, 5), np.float32), np.zeros(n_frames, np.int64)
    for t in range(n_frames):
        rows = []
        for p in range(4):
            if (t + 3 * p) % 9 == 0:
                continue                                   # missed detection
            x = 50 + 140 * p + (9.0 - 4.0 * p) * t         # persons 0 and 3 cross
            y = 40 + 12 * p
            rows.append([x, y, x + 80, y + 200, 0.45 + 0.13 * p + 0.01 * (t % 5)])     # some scores around obj_score_thr
        rows.append([rng.uniform(0, 600), rng.uniform(0, 300), 0, 0, 0.3])
        rows[-1][2], rows[-1][3] = rows[-1][0] + 40, rows[-1][1] + 90
        dets[t, :len(rows)] = rows
        counts[t] = len(rows)
    return {"dets": dets, "counts": counts}


def mmtrack_oracle(d, ref=None):
    from oracle.tracking import SortTrackerRef
    trk = SortTrackerRef()
    out = {}
    for t in range(len(d["counts"])):
        out[f"rows{t}"] = trk.step(d["dets"][t, : int(d["counts"][t])])
    return out


def mmtrack_checks(d):
    return {f"rows{t}": "exact" for t in range(len(d["counts"]))}


# ---- networks: mmpose HRNet + head, VideoPose3D TemporalModelOptimized1f, with the seeded state dicts --------------------------
NET_TOL = 2e-4      # of the output's range: cuDNN / MKL sum in other orders than the fmaf chain (tests/test_oracle_nets.py uses the same bar)


def nets_inputs(rng):
    return {"x_hrnet": rng.standard_normal((1, 3, 64, 64)).astype(np.float32),
            "kp_vp3d": (np.cumsum(rng.normal(0, 0.01, (30, 17, 2)), axis=0) + rng.uniform(-0.5, 0.5, (1, 17, 2))).astype(np.float32)}


def nets_state_dicts():
    from posepipeline_amd.models import hrnet, synth
    from posepipeline_amd.models import videopose3d as vp3d
    spec = hrnet.HRNetSpec(32, 17, 64, 64)
    return (spec, synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1),
            synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3))


def nets_oracle(d, ref=None):
    from oracle import nets as onets
    spec, hsd, vsd = nets_state_dicts()
    return {"hrnet_heatmaps": onets.HRNetRef(hsd, 32).forward(d["x_hrnet"]),
            "vp3d": onets.VideoPose3DRef(vsd).forward(onets.videopose3d_windows(d["kp_vp3d"], 121))}


def nets_checks(d):
    return {"hrnet_heatmaps": ("range", NET_TOL), "vp3d": ("range", NET_TOL)}


SECTIONS = {
    "cv2_affine": (cv2_affine_inputs, cv2_affine_oracle, cv2_affine_checks, ["cv2"]),
    "cv2_blur": (cv2_blur_inputs, cv2_blur_oracle, cv2_blur_checks, ["cv2"]),
    "cv2_resize": (cv2_resize_inputs, cv2_resize_oracle, cv2_resize_checks, ["cv2"]),
    "cv2_nv12": (cv2_nv12_inputs, cv2_nv12_oracle, cv2_nv12_checks, ["cv2"]),
    "mmpose": (mmpose_inputs, mmpose_oracle, mmpose_checks, ["mmpose", "cv2"]),
    "mmcv": (mmcv_inputs, mmcv_oracle, mmcv_checks, ["mmcv", "torch"]),
    "mmtrack": (mmtrack_inputs, mmtrack_oracle, mmtrack_checks, ["mmtrack", "torch"]),
    "nets": (nets_inputs, nets_oracle, nets_checks, ["mmpose", "torch"]),
}


def section_rng(name):
    return np.random.default_rng([SEED, sum(map(ord, name))])


def compare(name, fixture, got):
    """-> list of human-readable mismatches of oracle outputs `got` against the fixture's third-party outputs"""
    _, _, checks, _ = SECTIONS[name]
    bad = []
    for key, rule in checks(fixture).items():
        if key not in fixture:
            bad.append(f"{key}: missing from the fixture (the generator could not produce it)")
            continue
        a, b = np.asarray(fixture[key]), np.asarray(got[key])
        if a.shape != b.shape:
            bad.append(f"{key}: shape {b.shape} != fixture {a.shape}")
        elif rule == "exact":
            if not np.array_equal(a, b):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                bad.append(f"{key}: {int((d > 0).sum())} of {d.size} values differ, max abs {d.max():.3g}")
        else:
            tol = rule[1] * float(np.abs(a).max()) if isinstance(rule, tuple) else float(rule)
            d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
            if not d <= tol:
                bad.append(f"{key}: max abs diff {d:.3g} > {tol:.3g}")
    return bad
