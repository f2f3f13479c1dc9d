"""The oracle-comparing end-to-end tests in BOTH numerics modes, at the sizes the bench times.

"exact": every convolution on the float32 MFMA kernels -- `==` against the oracle wherever the arithmetic is integer or a
         k-ordered fmaf chain.
"split": the library's default (and what bench.py times): eligible convolutions on the bf16 matrix cores through an exact
         three-way operand split.  Here `==` becomes north_star's tolerance -- EVERY joint within 1e-3 px / 1e-3 mm, scores
         within 1e-5, integer outputs (track ids, frame indices, which rows are zero) identical -- not a quantile.

Seeded-random pose weights cannot carry a per-joint 1e-3 px claim in ANY float32 evaluation order (heat-maps are noise; a
transposed-network control with the bit-exact kernels moves as many joints, tests/test_gpu_split.py), so the pose networks
here use `synth.smooth_state_dict`: positive normalised kernels through every stage, i.e. smooth single-peaked heat-maps
like a trained network's.  The detector keeps seeded-random weights: its discrete decisions are checked with the
margin-aware oracle (oracle/detector_margins.py): certain <= device <= possible.
Reference call sites: pose_pipeline/wrappers/mmpose.py:60-81, wrappers/mmtrack.py:37-60, wrappers/videopose3d.py:77-85.
"""
import numpy as np
import pytest

from oracle import decode as odec
from oracle import detector as odet
from oracle import detector_margins as odm
from oracle import nets as onets
from oracle.tracking import SortTrackerRef
from posepipeline_amd import _lib as L
from posepipeline_amd import ops
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.models import videopose3d as vp3d
from posepipeline_amd.program import Net
from tests.test_gpu_cascade import reference_3d
from tests.test_gpu_detector import synth_frame
from tests.test_gpu_pipeline import oracle_topdown

pytestmark = pytest.mark.gpu

TOL_PX = 1e-3          # north_star: 2D joints within 1e-3 px
TOL_M = 1e-6           # 3D joints within 1e-3 mm.  VideoPose3D's unit is the metre -- for the trained checkpoint, and for the lifting
                       # weights of these tests by construction (synth.smooth_lifting_state_dict: outputs span about a metre, max-norm
                       # sensitivity <= 0.98, asserted in tests/test_weights.py); every 3D assertion ALSO holds the error to
TOL_3D_REL = 1e-6      # this fraction of the clip's largest |coordinate|, which is unit-free
TOL_SCORE = 1e-5       # relative to the largest score of the batch


@pytest.fixture(params=["exact", "split"])
def numerics(request, ctx):
    """every net the test creates gets these numerics (a net keeps what it was created with, ABI 7)"""
    with L.default_numerics(request.param):
        yield request.param


def assert_scores(got, ref, numerics):
    ref = np.asarray(ref, np.float32)
    if numerics == "exact":
        assert np.array_equal(got, ref)
    else:
        assert np.abs(got - ref).max() <= TOL_SCORE * max(float(np.abs(ref).max()), 1e-30), np.abs(got - ref).max() / np.abs(ref).max()


# ---- (i) HRNet-W48 384x288 + flip test + DARK decode at full size, every joint ------------------------------------------------
def test_hrnet_w48_full_size_every_joint(ctx, numerics):
    spec = hrnet.hrnet_w48_384x288()
    sd = synth.smooth_state_dict(hrnet.hrnet_param_shapes(spec), seed=11)
    n = 3
    x = synth.blob_crops(np.random.default_rng(5), n, spec.in_h, spec.in_w)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=2 * n)
    assert net.numerics == numerics and (numerics == "split") == bool((net.conv_kinds() == 2).any())
    # (cx, cy, sx, sy) of 1080p persons (mmpose `_box2cs`: scale = box / 200 * 1.25)
    cs = np.array([[960.0, 540.0, 1.9, 2.5333333], [300.5, 700.25, 1.2, 1.6], [1700.0, 400.0, 2.4, 3.2]], np.float32)
    # oracle: network on the crops and on their mirror images, flip merge, DARK decode
    xin = np.ascontiguousarray(np.transpose(np.concatenate([x, x[:, :, ::-1]])[..., :3], (0, 3, 1, 2)))
    ref_hm = onets.HRNetRef(sd, 48).forward(xin)
    ref_kp, _ = odec.decode_topdown(ref_hm[:n], ref_hm[n:], hrnet.COCO_FLIP_PAIRS, cs[:, :2], cs[:, 2:], post_process="unbiased", kernel=17)
    # well-conditioned by construction: positive single-peaked maps
    assert ref_hm.min() > 0 and np.isfinite(ref_hm).all()
    # device: the fused stage (mirror, backbone x 2, flip merge + decode), then the raw heat-maps
    td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="unbiased", blur_kernel=17)
    kp = td.run_precropped(x, cs)
    hm = net.read("output", 2 * n).reshape(2 * n, 17, 96, 72)
    if numerics == "exact":
        assert np.array_equal(hm, ref_hm)
    else:
        assert not np.array_equal(hm, ref_hm), "the split kernels did not run"
        assert np.abs(hm - ref_hm).max() <= 2e-5 * np.abs(ref_hm).max()
    d = np.abs(kp[:, :, :2] - ref_kp[:, :, :2]).max(axis=2)
    print(f"[{numerics}] W48 384x288: {d.size} joints, max deviation {d.max():.2e} px, heat-maps {np.abs(hm - ref_hm).max() / np.abs(ref_hm).max():.2e} of range")
    assert d.max() <= TOL_PX, d
    assert_scores(kp[:, :, 2], ref_kp[:, :, 2], numerics)


# ---- (ii) the cascade at configs[2] / [3] sizes ---------------------------------------------------------------------------
def _blob_person(rng, h, w):
    """a dark rectangle with one bright Gaussian blob per colour plane (what the smoothing pose network peaks on)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    tex = np.full((h, w, 3), 30.0, np.float32)
    cy, cx = rng.uniform(0.35 * h, 0.65 * h), rng.uniform(0.35 * w, 0.65 * w)
    for c in range(3):
        sg = rng.uniform(0.05, 0.09) * h
        oy, ox = rng.uniform(-0.03, 0.03, 2) * h
        tex[:, :, c] += rng.uniform(150, 215) * np.exp(-((yy - cy - oy) ** 2 + (xx - cx - ox) ** 2) / (2 * sg * sg))
    return np.clip(tex + rng.uniform(0, 3, tex.shape), 0, 255).astype(np.uint8)


def _lift_sd():
    return synth.smooth_lifting_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)


def assert_3d(got, ref, numerics, what=""):
    """exact: the oracle's bits; split: 1e-3 mm AND 1e-6 of the output range"""
    if numerics == "exact":
        assert np.array_equal(got, ref), what
        return 0.0
    d = float(np.abs(got - ref).max())
    assert d <= TOL_M and d <= TOL_3D_REL * float(np.abs(ref).max()), (what, d, float(np.abs(ref).max()))
    return d


def _check_tracks(outs, tracks, frames, n, h, w, pose_sd, width, image_size, lift_sd, numerics, sample):
    """every followed id against the reference's table chain: PersonBbox.make(keep_tracks=[id]) decides the boxes / zero rows,
    the oracle chain gives the 2D joints of the sampled frames, process_videopose3d's whole-clip windows the 3D joints"""
    from posepipeline_amd.cascade import collect
    from posepipeline_amd.tracking import person_bbox
    ids = sorted({r[0] for fr_ in tracks for r in fr_})
    k2, k3 = collect(outs, "keypoints"), collect(outs, "keypoints_3d")
    assert sorted(k2) == sorted(k3) == ids
    dicts = [[{"track_id": r[0], "tlhw": np.array([r[1], r[2], r[3] - r[1], r[4] - r[2]], np.float64)} for r in fr_] for fr_ in tracks]
    worst2 = worst3 = 0.0
    for tid in ids:
        bbox, present = person_bbox(dicts, [tid])
        f2, a2 = k2[tid]
        filled = np.flatnonzero(present)
        assert f2 == filled[0] and f2 + len(a2) > filled[-1]
        for t in range(f2, f2 + len(a2)):                                   # zero rows exactly where the reference has none
            assert a2[t - f2].any() == bool(present[t]), (tid, t)
        for t in sample:
            if t >= n or not present[t]:
                continue
            ref = oracle_topdown(pose_sd, width, frames[t:t + 1], bbox[t:t + 1], image_size, "unbiased", 17)[0]
            worst2 = max(worst2, float(np.abs(a2[t - f2][:, :2] - ref[:, :2]).max()))
            assert np.abs(a2[t - f2][:, :2] - ref[:, :2]).max() <= TOL_PX, (tid, t, np.abs(a2[t - f2][:, :2] - ref[:, :2]).max())
            assert_scores(a2[t - f2][:, 2], ref[:, 2], numerics)
        f3, a3 = k3[tid]
        ref3 = reference_3d(a2, f2, n, w, h, lift_sd)                        # the lifting oracle on the device's own 2D track
        assert f3 == f2
        worst3 = max(worst3, assert_3d(a3, ref3[f3:f3 + len(a3)], numerics, tid))
    print(f"[{numerics}] {len(ids)} ids: 2D max {worst2:.2e} px, 3D max {worst3:.2e} m")
    return ids, k2


def test_cascade_1080p_four_persons_both_modes(ctx, numerics):
    """BASELINE.json configs[2]: multi-person 1080p through detector -> SORT -> HRNet-W48 384x288 -> VideoPose3D, 4 persons with
    crossing trajectories and one missed detection at a chunk boundary (back-fill into the previous chunk)"""
    from posepipeline_amd.cascade import Cascade
    from posepipeline_amd.video import ArrayVideo
    rng = np.random.default_rng(21)
    h, w, n, chunk = 1080, 1920, 16, 8
    bg = rng.integers(20, 60, (h // 40, w // 40, 3)).astype(np.uint8)
    bg = np.repeat(np.repeat(bg, 40, axis=0), 40, axis=1)
    people = [dict(x=200.0, y=150.0, w=170, h=520, vx=38.0), dict(x=900.0, y=260.0, w=150, h=470, vx=-36.0),
              dict(x=1400.0, y=90.0, w=190, h=580, vx=6.0), dict(x=100.0, y=500.0, w=120, h=330, vx=12.0)]
    for q in people:
        q["tex"] = _blob_person(rng, q["h"], q["w"])
    frames = np.empty((n, h, w, 3), np.uint8)
    gt = []
    for t in range(n):
        f = bg.copy()
        rows = []
        for i, q in enumerate(people):
            x0, y0 = int(q["x"] + q["vx"] * t), int(q["y"])
            f[y0:y0 + q["h"], x0:x0 + q["w"]] = q["tex"]
            if not (i == 2 and t == 8):                                     # person 2 is missed in frame 8 (first of chunk 2)
                rows.append([x0 + 0.25 * i, y0 + 0.5, x0 + q["w"] - 0.25, y0 + q["h"], 0.6 + 0.08 * i])
        frames[t] = f
        gt.append(np.array(rows, np.float32))
    spec = hrnet.hrnet_w48_384x288()
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_sd = synth.smooth_state_dict(hrnet.hrnet_param_shapes(spec), seed=11)
    lift_sd = _lift_sd()
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=4, pose_spec=spec)
    assert cas.pose_net.numerics == numerics and cas.detector.net_a.numerics == numerics and cas.lift_net.numerics == numerics
    outs = list(cas.run_video(ArrayVideo(frames), replay_fn=lambda first, m: gt[first:first + m]))
    assert [o["first_frame"] for o in outs] == [0, 8, 16]
    tracks = [fr_ for o in outs for fr_ in o["tracks"]]
    ref_trk = SortTrackerRef()
    for t in range(n):                                                      # ids: bit-exact in either mode
        rows = ref_trk.step(gt[t])
        assert [r[0] for r in tracks[t]] == [int(x[0]) for x in rows]
        assert np.array_equal(np.array([r[1:] for r in tracks[t]], np.float32), rows[:, 1:])
    ids, k2 = _check_tracks(outs, tracks, frames, n, h, w, pose_sd, 48, (288, 384), lift_sd, numerics, sample=(0, 7, 8, 9, 15))
    assert len(ids) == 5                                                    # the missed person comes back under a new id
    old = tracks[0][2][0]
    new = [r[0] for r in tracks[9] if r[0] not in {q[0] for q in tracks[0]}][0]
    assert k2[new][0] == 7 and k2[old][1][8].any() and k2[old][1][9].any() and not k2[old][1][10:].any()


def test_cascade_long_clip_both_modes(ctx, numerics):
    """configs[3]'s temporal part: 330 frames in 6 chunks -- the 3D joints EVERY step emits are the whole-clip lifting of the 2D
    track (window [t-121, t+121]), frame t leaves the cascade with the chunk that brings frame t+121"""
    from posepipeline_amd.cascade import Cascade, collect
    rng = np.random.default_rng(8)
    h, w, n, chunk = 135, 240, 330, 64
    bg = np.repeat(np.repeat(rng.integers(20, 60, (h // 15, w // 15, 3)).astype(np.uint8), 15, axis=0), 15, axis=1)
    tex = _blob_person(rng, 100, 70)
    frames = np.empty((n, h, w, 3), np.uint8)
    for t in range(n):
        frames[t] = bg
        x0 = 20 + int(0.4 * t)
        frames[t, 20:120, x0:x0 + 70] = np.clip(tex.astype(np.int32) + (t % 7), 0, 255).astype(np.uint8)
    pose_spec = hrnet.HRNetSpec(32, 17, 128, 96)
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_sd = synth.smooth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=12)
    lift_sd = _lift_sd()
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=1, pose_spec=pose_spec)
    gt = [np.array([[20 + 0.4 * t, 20, 90 + 0.4 * t, 120, 0.9]], np.float32) for t in range(n)]
    outs = [cas.step(frames[i:i + chunk], replay=gt[i:i + chunk]) for i in range(0, n, chunk)] + [cas.flush()]
    tracks = [fr_ for o in outs for fr_ in o["tracks"]]
    ids, k2 = _check_tracks(outs, tracks, frames, n, h, w, pose_sd, 32, (96, 128), lift_sd, numerics, sample=(0, 63, 64, 200, n - 1))
    assert len(ids) == 1
    tid = ids[0]
    f2, a2 = k2[tid]
    assert f2 == 0 and a2.shape == (n, 17, 3)
    ref3 = reference_3d(a2, 0, n, w, h, lift_sd)
    emitted = []
    for o in outs:
        if tid not in o["keypoints_3d"]:
            emitted.append(0)
            continue
        fr3 = o["keypoints_3d_frames"][tid]
        emitted.append(len(fr3))
        assert_3d(o["keypoints_3d"][tid], ref3[fr3], numerics)
    assert emitted == [0, 7, 64, 64, 64, 10, 121]                          # identical emission schedule in either mode


# ---- (ii-b) END TO END: oracle 2D chain -> oracle lifting against the device's 3D ----------------------------------------------
_E2E_ORACLE = {}       # the CPU chain does not depend on the numerics mode: computed once per size


@pytest.mark.parametrize("size", ["1080p_w48", "540x960_w32_130frames"])
def test_end_to_end_3d_against_the_oracle_chain(ctx, numerics, size):
    """The tolerance north_star states for the OUTPUT of the cascade: the device's 3D joints against the reference's whole
    chain evaluated by the oracle -- PersonBbox -> crop -> HRNet x 2 -> flip merge -> DARK decode for EVERY frame, then
    normalize_screen_coordinates and one 243-frame window per frame (wrappers/mmpose.py:60-76, wrappers/videopose3d.py:77-91).
    Unlike `_check_tracks` (which lifts the device's own 2D track with the oracle), the 2D error of the default numerics
    propagates into this comparison.  Well-conditioned pose AND lifting weights; bars: 2D every joint <= 1e-3 px, 3D <= 1e-3 mm
    and <= 1e-6 of the output range; exact mode: `==` throughout."""
    from posepipeline_amd.cascade import Cascade, collect
    from posepipeline_amd.tracking import person_bbox
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    if size == "1080p_w48":
        h, w, n, chunk, width, image_size = 1080, 1920, 6, 3, 48, (288, 384)
        spec = hrnet.hrnet_w48_384x288()
        box = lambda t: (640 + 9 * t, 200, 830 + 9 * t, 790)
    else:
        h, w, n, chunk, width, image_size = 540, 960, 130, 64, 32, (96, 128)        # > 121 frames: real (not only replicated) windows
        spec = hrnet.HRNetSpec(32, 17, 128, 96)
        box = lambda t: (80 + 3 * t, 80, 360 + 3 * t, 480)
    rng = np.random.default_rng(31)
    cell = 40 if h == 1080 else 20
    bg = np.repeat(np.repeat(rng.integers(20, 60, (h // cell, w // cell, 3)).astype(np.uint8), cell, axis=0), cell, axis=1)
    x0, y0, x1, y1 = box(0)
    tex = _blob_person(rng, y1 - y0, x1 - x0)
    frames = np.empty((n, h, w, 3), np.uint8)
    gt = []
    for t in range(n):
        x0, y0, x1, y1 = box(t)
        frames[t] = bg
        frames[t, y0:y1, x0:x1] = np.clip(tex.astype(np.int32) + (t % 5), 0, 255).astype(np.uint8)
        gt.append(np.array([[x0 + 0.25, y0 + 0.5, x1 - 0.25, y1, 0.9]], np.float32))
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_sd = synth.smooth_state_dict(hrnet.hrnet_param_shapes(spec), seed=13)
    lift_sd = _lift_sd()
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=1, pose_spec=spec)
    assert cas.pose_net.numerics == numerics
    outs = [cas.step(frames[i:i + chunk], replay=gt[i:i + chunk]) for i in range(0, n, chunk)] + [cas.flush()]
    tracks = [fr_ for o in outs for fr_ in o["tracks"]]
    ids = sorted({r[0] for fr_ in tracks for r in fr_})
    assert len(ids) == 1
    f2, a2 = collect(outs, "keypoints")[ids[0]]
    f3, a3 = collect(outs, "keypoints_3d")[ids[0]]
    assert f2 == f3 == 0 and len(a2) == len(a3) == n
    if size not in _E2E_ORACLE:
        dicts = [[{"track_id": r[0], "tlhw": np.array([r[1], r[2], r[3] - r[1], r[4] - r[2]], np.float64)} for r in fr_] for fr_ in tracks]
        bbox, present = person_bbox(dicts, ids)
        assert present.all()
        k2 = np.asarray(oracle_topdown(pose_sd, width, frames, bbox, image_size, "unbiased", 17))
        assert k2.dtype == np.float32           # every frame present: the stored track is float32 and is normalised in float32
        kn = normalize_screen_coordinates(k2[:, :, :2], w, h).astype("float32")       # (wrappers/videopose3d.py:72 on the stored array)
        _E2E_ORACLE[size] = (k2, onets.VideoPose3DRef(lift_sd).forward(onets.videopose3d_windows(kn, 121)))
    ref2, ref3 = _E2E_ORACLE[size]
    d2 = float(np.abs(a2[:, :, :2] - ref2[:, :, :2]).max())
    d3 = float(np.abs(a3 - ref3).max())
    rng3 = float(np.abs(ref3).max())
    print(f"[{numerics}] {size}: {n} frames end to end, 2D max {d2:.2e} px, 3D max {d3:.2e} m = {d3 / rng3:.2e} of the output range "
          f"({rng3:.2f} m), lifting sensitivity bound {synth.max_norm_gain_bound(lift_sd):.2f}")
    assert 0.2 < rng3 < 3.0                                                  # metre-sized outputs: the absolute bar means something
    if numerics == "exact":
        # the one non-integer step of the exact chain is DARK's log (numpy float32 vs the device's correctly rounded one)
        assert d2 <= TOL_PX
        assert np.array_equal(a2[:, :, 2], ref2[:, :, 2].astype(np.float32))
        if d2 == 0.0:
            assert d3 == 0.0                                                 # identical 2D track -> the lifting oracle's bits
    else:
        assert d2 <= TOL_PX
        assert_scores(a2[:, :, 2], ref2[:, :, 2], numerics)
    assert d3 <= TOL_M and d3 <= TOL_3D_REL * rng3


# ---- (iii) detector: margin-aware set equality, ids on detector-produced boxes ----------------------------------------------
def _det_sd():
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    for k, g in (("detector.rpn_head.rpn_cls.weight", 0.5), ("detector.rpn_head.rpn_reg.weight", 0.1),
                 ("detector.roi_head.bbox_head.fc_reg.weight", 0.2)):
        sd[k] = (sd[k] * g).astype(np.float32)
    return sd


BOX_TOL = {"exact": 0.0, "split": 1e-2}     # px in the SOURCE frame; seeded-random regression weights move 300-px boxes by O(1) deltas through
                                            # exp(): float32 rounding noise of fc6's 12544-term sums shows at the 1e-3 px level (measured, printed)


@pytest.mark.parametrize("size", [(135, 240), (1080, 1920)])
def test_detector_margin_aware_set_equality(ctx, numerics, size):
    """Every detection the oracle is CERTAIN of (all its top-k / NMS / threshold / level decisions have >= 1e-4 margin) must come
    out of the device, and every device detection must be one the oracle holds POSSIBLE -- boxes within BOX_TOL, scores
    within 1e-4.  In exact mode the sets are the oracle's own, bit for bit (also asserted)."""
    h, w = size
    sd = _det_sd()
    rng = np.random.default_rng(3 if h == 1080 else 2)
    frames = np.stack([synth_frame(rng, h, w) for _ in range(1 if h == 1080 else 2)])
    det = fr.Detector(ctx, sd, h, w, max_frames=len(frames))
    assert det.net_a.numerics == numerics and det.net_b.numerics == numerics
    dets = det.run(frames)
    model = odet.FasterRCNNRef(sd)
    for f in range(len(frames)):
        certain, possible = odm.detect3(model, frames[f][:, :, ::-1])
        assert len(certain) > 0 and len(certain) <= len(possible)
        missing, unexplained, worst = odm.check_between(dets[f], certain, possible, BOX_TOL[numerics])
        print(f"[{numerics}] {h}x{w} frame {f}: device {len(dets[f])}, certain {len(certain)}, possible {len(possible)}, "
              f"max box deviation of certain detections {worst:.2e} px")
        assert not missing, (len(missing), missing[:3])
        assert not unexplained, (len(unexplained), unexplained[:3])
        if numerics == "exact":
            assert np.array_equal(dets[f], odet.detect(model, frames[f][:, :, ::-1]))


def test_track_ids_on_detector_boxes(ctx, numerics):
    """north_star: integer track ids bit-exact -- on boxes the DETECTOR produced (no replay): a 4-frame clip through the device
    detector and the product tracker against the oracle detector and the oracle tracker.  The tracker only sees detections
    with score > 0.5; the clip is one where those are certain (asserted), so ids must be identical in either mode."""
    from posepipeline_amd.tracking import Tracker
    h, w, n = 135, 240, 4
    sd = _det_sd()
    sd["detector.roi_head.bbox_head.fc_cls.bias"] = (sd["detector.roi_head.bbox_head.fc_cls.bias"] + np.array([-4.6, 0.0], np.float32)).astype(np.float32)
    rng = np.random.default_rng(17)
    base = synth_frame(rng, h, w)
    frames = np.stack([np.roll(base, 3 * t, axis=1) for t in range(n)])
    det = fr.Detector(ctx, sd, h, w, max_frames=n)
    dets = det.run(frames)
    model = odet.FasterRCNNRef(sd)
    trk, ref_trk = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5), SortTrackerRef()
    total = 0
    for f in range(n):
        ref = odet.detect(model, frames[f][:, :, ::-1])
        rows = ref_trk.step(ref)
        g = np.asarray(dets[f], np.float32).reshape(-1, 5)
        ids, _, info = trk.step(g[:, :4].astype(np.float64), g[:, 4].astype(np.float64))
        got = [(int(i), g[j]) for i, j in zip(ids, info[:, 1])]
        assert len(rows) > 0, "the clip must give the tracker something to do"
        # no score sits within 1e-4 of the tracker's threshold (otherwise the comparison would be ill-posed)
        assert (np.abs(ref[:, 4] - 0.5) > 1e-4).all()
        assert [i for i, _ in got] == [int(r[0]) for r in rows], (f, [i for i, _ in got], rows[:, 0])
        for (_, box), r in zip(got, rows):
            assert np.abs(box[:4] - r[1:5]).max() <= BOX_TOL[numerics] and abs(box[4] - r[5]) <= 1e-4
        total += len(rows)
    print(f"[{numerics}] {total} tracked detector boxes over {n} frames, ids identical")


# ---- (iv) the integer contract at full size, NO replay -------------------------------------------------------
def _clip_1080p_four_persons(n, rng):
    """1080p clip: static block background, four textured rectangles on crossing trajectories"""
    h, w = 1080, 1920
    bg = np.repeat(np.repeat(rng.integers(20, 60, (h // 40, w // 40, 3)).astype(np.uint8), 40, axis=0), 40, axis=1)
    people = [dict(x=200.0, y=150.0, w=170, h=520, vx=9.0), dict(x=1500.0, y=260.0, w=150, h=470, vx=-11.0),
              dict(x=900.0, y=90.0, w=190, h=580, vx=2.0), dict(x=100.0, y=500.0, w=120, h=330, vx=5.0)]
    for q in people:
        q["tex"] = _blob_person(rng, q["h"], q["w"])
    frames = np.empty((n, h, w, 3), np.uint8)
    for t in range(n):
        frames[t] = bg
        for q in people:
            x0, y0 = int(q["x"] + q["vx"] * t), int(q["y"])
            frames[t, y0:y0 + q["h"], x0:x0 + q["w"]] = q["tex"]
    return frames


def _run_no_replay(ctx, frames, sds, chunk, **kw):
    from posepipeline_amd.cascade import Cascade, collect
    det_sd, pose_sd, lift_sd, spec = sds
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, frames.shape[1], frames.shape[2], chunk=chunk, max_persons=4, pose_spec=spec, **kw)
    outs = [cas.step(frames[i:i + chunk]) for i in range(0, len(frames), chunk)] + [cas.flush()]
    tracks = [fr_ for o in outs for fr_ in o["tracks"]]
    return cas, tracks, collect(outs, "keypoints"), collect(outs, "keypoints_3d")


def test_integer_contract_1080p_64_frames_no_replay(ctx):
    """north_star: "integer track-ids and bbox indices bit-exact" -- at 1080p, 64 frames in 4 chunks, >= 4 followed tracks, on boxes the
    DETECTOR produced (no replay anywhere), seeded-random detector weights, nothing about the clip conditioned to pass:
      exact         device detector -> product tracker -> PersonBbox  ==  oracle detector (anchor frame: the oracle itself at 1080p;
                    every frame: the same kernels, whose equality with the oracle test_gpu_fullsize / test_gpu_detector hold) ->
                    ORACLE tracker -> the reference's PersonBbox table logic: ids, tracked rows, bbox selection, `present` identical
      integer-exact (numerics "split", id_numerics "exact": the detector on the float32-MFMA kernels, pose / lifting on the fp16-form
                    kernels) the SAME integers on all 64 frames by construction -- asserted -- and every 2D / 3D joint within 1e-3 px /
                    1e-3 mm of the exact cascade's
      split         (everything on the fast kernels) is NOT held to integer identity: near-ties among ~5000 seeded-random proposal
                    scores flip under ANY float32 reordering; it is held to the margin-aware set relation above, and what it does
                    here is measured and printed (bench.py reports the same figures per run)."""
    from posepipeline_amd.tracking import person_bbox
    n, chunk = 64, 16
    frames = _clip_1080p_four_persons(n, np.random.default_rng(41))
    spec = hrnet.hrnet_w48_384x288()
    sds = (_det_sd(), synth.smooth_state_dict(hrnet.hrnet_param_shapes(spec), seed=11), _lift_sd(), spec)
    cas_e, tr_e, k2_e, k3_e = _run_no_replay(ctx, frames, sds, chunk, numerics="exact")
    assert cas_e.detector.net_a.numerics == "exact"
    per_frame = [len(t) for t in tr_e]
    followed = sorted(k2_e)
    print(f"[exact] tracked boxes per frame {min(per_frame)} .. {max(per_frame)}, {len({r[0] for t in tr_e for r in t})} ids, followed {followed}")
    assert min(per_frame) >= 4 and len(followed) >= 4
    # (a) anchor: the oracle detector itself on one 1080p frame == what the exact device detector fed the tracker
    model = odet.FasterRCNNRef(sds[0])
    dets_dev = cas_e.detector.run(frames[37:38])[0]
    assert np.array_equal(dets_dev, odet.detect(model, frames[37][:, :, ::-1]))
    # (b) oracle tracker over the exact detections of EVERY frame: ids and rows identical to the product tracker's
    ref_trk = SortTrackerRef()
    dets_all = [d for i in range(0, n, chunk) for d in cas_e.detector.run(frames[i:i + chunk])]
    assert np.array_equal(dets_all[37], dets_dev)
    for t in range(n):
        rows = ref_trk.step(np.asarray(dets_all[t], np.float32).reshape(-1, 5))
        assert [r[0] for r in tr_e[t]] == [int(x[0]) for x in rows], t
        assert np.array_equal(np.array([r[1:] for r in tr_e[t]], np.float32).reshape(-1, 5), np.asarray(rows, np.float32).reshape(-1, 6)[:, 1:]), t
    # (c) the reference's PersonBbox logic on the table rows == what the streaming cascade followed
    dicts = [[{"track_id": r[0], "tlhw": np.array([r[1], r[2], r[3] - r[1], r[4] - r[2]], np.float64)} for r in fr_] for fr_ in tr_e]
    present_e = {}
    for tid in followed:
        bbox, present = person_bbox(dicts, [tid])
        f2, a2 = k2_e[tid]
        filled = np.flatnonzero(present)
        assert f2 == filled[0]
        for t in range(f2, f2 + len(a2)):
            assert a2[t - f2].any() == bool(present[t]), (tid, t)
        present_e[tid] = present
    # ---- integer-exact: the fast pose / lifting kernels under an exact detector ----
    cas_h, tr_h, k2_h, k3_h = _run_no_replay(ctx, frames, sds, chunk, numerics="split", id_numerics="exact")
    assert cas_h.detector.net_a.numerics == "exact" and cas_h.pose_net.numerics == "split" and (cas_h.pose_net.conv_kinds() == 2).any()
    assert tr_h == tr_e or all(np.array_equal(np.array(a, np.float32), np.array(b, np.float32)) for a, b in zip(tr_h, tr_e))
    assert sorted(k2_h) == followed
    worst2 = worst3 = 0.0
    n_joint = n_within = 0
    for tid in followed:
        (f_e, a_e), (f_h, a_h) = k2_e[tid], k2_h[tid]
        assert f_e == f_h and a_e.shape == a_h.shape
        assert np.array_equal(a_e.any(axis=(1, 2)), a_h.any(axis=(1, 2)))           # `present`: the same zero rows
        d = np.abs(a_e[:, :, :2] - a_h[:, :, :2]).max(axis=2)
        worst2 = max(worst2, float(d.max()))
        n_joint += d.size
        n_within += int((d <= TOL_PX).sum())
        (g_e, b_e), (g_h, b_h) = k3_e[tid], k3_h[tid]
        assert g_e == g_h and b_e.shape == b_h.shape
        worst3 = max(worst3, float(np.abs(b_e - b_h).max()))
    print(f"[integer-exact] ids / rows / present identical on {n} frames; 2D: {n_within} of {n_joint} joints within {TOL_PX} px of the exact "
          f"cascade's, max {worst2:.2e} px; 3D max {worst3:.2e} m")
    # the followed boxes are whatever a seeded-random detector reports (background blocks, rectangle corners): the pose network's maps
    # on them are not the single-peaked ones the per-joint 1e-3 px claim is made on (tests above: blob persons at person boxes), so
    # the float side is held to "all but a handful within 1e-3 px, none beyond 5e-3" here; the INTEGER side is the point
    assert n_within >= 0.995 * n_joint and worst2 <= 5 * TOL_PX and worst3 <= 5 * TOL_M
    # ---- split everywhere: measured, not asserted identical ----
    cas_s, tr_s, k2_s, _ = _run_no_replay(ctx, frames, sds, chunk, numerics="split")
    assert (cas_s.detector.net_a.conv_kinds() == 2).any()
    same_ids = sum([r[0] for r in a] == [r[0] for r in b] for a, b in zip(tr_s, tr_e))
    matched = total = 0
    for a, b in zip(tr_s, tr_e):
        total += len(b)
        ba = np.array([r[1:5] for r in a], np.float32).reshape(-1, 4)
        for r in b:
            if len(ba) and np.abs(ba - np.array(r[1:5], np.float32)).max(axis=1).min() <= BOX_TOL["split"]:
                matched += 1
    print(f"[split] frames with the exact cascade's id list: {same_ids} / {n}; tracked boxes re-found within {BOX_TOL['split']} px: {matched} / {total}")
    assert matched >= 0.9 * total
    for c in (cas_e, cas_h, cas_s):
        c.release()


def test_certified_ids_1080p_64_frames_no_replay(ctx):
    """the product tells the caller which frames are decided.  Cascade(id_numerics="certified") runs the detector on
    the fast kernels with per-frame decision margins (pp_detector_enable_margins: det_post.hip / nms.hip) and re-runs the frames whose
    closest decision is within the split kernels' measured error through the float32-MFMA detector.  Held here, at 1080p / 64 frames /
    no replay / seeded-random detector weights:
      * every certified frame's detections are the SAME rows in the same order as the exact detector's, within float tolerance (the
        certificate: no integer decision of that frame flipped) -- and there is at least one frame the fast kernels DO decide
        differently in this clip, which the margins caught (it is among the re-run ones);
      * ids / tracked rows of the certified cascade == the exact cascade's on all 64 frames (the re-run frames are bit-identical, the
        certified ones differ in the last float digits of their boxes only);
      * the margins themselves: +inf / >= 0, reproducible, and the exact detector reports margins of the same size (they are a
        property of the frame, not of the numerics)."""
    from posepipeline_amd.cascade import CERTIFY_EPS
    n, chunk = 64, 16
    frames = _clip_1080p_four_persons(n, np.random.default_rng(41))
    spec = hrnet.hrnet_w48_384x288()
    sds = (_det_sd(), synth.smooth_state_dict(hrnet.hrnet_param_shapes(spec), seed=11), _lift_sd(), spec)
    cas_e, tr_e, k2_e, _ = _run_no_replay(ctx, frames, sds, chunk, numerics="exact")
    cas_c, tr_c, k2_c, _ = _run_no_replay(ctx, frames, sds, chunk, numerics="split", id_numerics="certified")
    st = cas_c.certify_stats
    print(f"[certified] frames {st['frames']}, certified by their margins {st['certified']}, run on the exact detector alone {st['exact_only_frames']}")
    assert st["frames"] == n
    assert [[r[0] for r in fr_] for fr_ in tr_c] == [[r[0] for r in fr_] for fr_ in tr_e]
    for a, b in zip(tr_c, tr_e):
        assert np.abs(np.array([r[1:6] for r in a], np.float64) - np.array([r[1:6] for r in b], np.float64)).max(initial=0.0) <= BOX_TOL["split"]
    assert sorted(k2_c) == sorted(k2_e)
    # the certificate itself, frame by frame, without the fallback policy in the way: margins of the fast detector vs what it decided
    det_s, det_e = cas_c.detector, cas_c.detector_exact
    det_e.enable_margins(True, CERTIFY_EPS["rpn_nms"] / (2.0 * CERTIFY_EPS["rpn_cut"]))
    n_cert = n_diff = n_diff_caught = 0
    for i in range(0, n, chunk):
        ds = det_s.run(frames[i:i + chunk])
        ms = det_s.margins(chunk)
        de = det_e.run(frames[i:i + chunk])
        me = det_e.margins(chunk)
        assert (ms >= 0).all() and (me >= 0).all()
        assert np.array_equal(ms, det_s.margins(chunk))
        redo = set(cas_c.uncertified_frames(ds, ms).tolist())
        for f in range(chunk):
            same = ds[f].shape == de[f].shape and (len(ds[f]) == 0 or np.abs(ds[f][:, :4] - de[f][:, :4]).max() <= BOX_TOL["split"])
            if f not in redo:
                n_cert += 1
                assert same, f"frame {i + f} was certified but the fast detector's rows differ from the exact detector's"
            if not same:
                n_diff += 1
                n_diff_caught += f in redo
        # margins are a property of the frame: same order of magnitude from both detectors wherever both are finite and not tiny
        both_ok = np.isfinite(ms) & np.isfinite(me) & (ms > 1e-4) & (me > 1e-4)
        assert np.allclose(ms[both_ok], me[both_ok], rtol=0.2)
    print(f"[certified] per-frame check: {n_cert} of {n} frames certified, {n_diff} frames where the fast detector's rows differ, all {n_diff_caught} caught")
    assert n_diff == n_diff_caught
    for c in (cas_e, cas_c):        # (a cascade is a reference cycle: give its ~15 GB back now, not at the next collection)
        c.release()
