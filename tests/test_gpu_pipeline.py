"""GPU parity of the remaining stages and of the drop-in boundary: 3D lifting, NMS, the fused top-down
stage, and the wrappers driven through the table shim (BASELINE.json configs[0] shape: a 32-frame
640x480 clip via TopDownPerson.populate())."""
import datetime
import os

import numpy as np
import pytest

from oracle import boxes as obox
from oracle import decode as odec
from oracle import nets as onets
from oracle import preprocess as opre
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.models import videopose3d as vp3d
from posepipeline_amd.program import Net

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- VideoPose3D: dilated whole-clip program == strided per-window oracle, bit for bit ----------------
@pytest.mark.parametrize("channels,n_frames,chunk", [(1024, 40, 512), (128, 300, 128), (128, 1, 128)])
def test_videopose3d_lift_bit_exact(ctx, channels, n_frames, chunk):
    from posepipeline_amd.wrappers.videopose3d import lift
    spec = vp3d.VideoPose3DSpec(channels=channels, chunk=chunk)
    sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(spec), seed=3)
    net = Net(ctx, vp3d.build_videopose3d_program(spec, sd), max_batch=2)
    rng = np.random.default_rng(n_frames)
    kp = np.cumsum(rng.normal(0, 0.01, (n_frames, 17, 2)), axis=0).astype(np.float32) + rng.uniform(-0.5, 0.5, (1, 17, 2)).astype(np.float32)
    got = lift(net, spec, kp)
    ref = onets.VideoPose3DRef(sd).forward(onets.videopose3d_windows(kp, spec.pad))
    assert got.shape == ref.shape == (n_frames, 17, 3)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-4
    assert np.array_equal(got, ref), np.abs(got - ref).max()


# ---- NMS ---------------------------------------------------------------------------------------------
def test_nms_mmcv_convention(ctx):
    rng = np.random.default_rng(9)
    for n in (1, 5, 64, 65, 700, 3000):
        c = rng.uniform(0, 1000, (max(n // 6, 1), 2))
        ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 12, (n, 2))
        wh = rng.uniform(20, 160, (n, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        for thr in (0.5, 0.7):
            keep = ops.nms(ctx, boxes, scores, thr, convention=0)
            assert keep.tolist() == obox.nms_mmcv(boxes, scores, thr), (n, thr)   # bit-exact indices, same order


def test_nms_deepsort_convention_golden(ctx):
    g = np.load(os.path.join(G, "nms_deepsort.npz"))
    checked = 0
    for k in range(int(g["n_cases"])):
        if not bool(g[f"n{k}_use_scores"]):
            continue                       # score-less calls sort by y2: not on the path (parser.py:66-70 passes scores)
        boxes, scores = g[f"n{k}_boxes"], g[f"n{k}_scores"]
        if len(np.unique(scores)) != len(scores):
            continue                       # np.argsort's order among equal scores is unspecified
        keep = ops.nms(ctx, boxes, scores, float(g[f"n{k}_thr"]), convention=1)
        assert keep.tolist() == g[f"n{k}_pick"].tolist(), k
        checked += 1
    assert checked >= 4


# ---- fused top-down stage == crop oracle -> network oracle -> decode oracle ------------------------------
def synth_clip(rng, n, h, w):
    """C1-style clip: noise background + one textured 'person' blob moving +3 px/frame."""
    frames = rng.integers(0, 256, (n, h, w, 3)).astype(np.uint8)
    boxes = []
    for t in range(n):
        x0, y0, bw, bh = w // 6 + 3 * t, h // 5, 3 * w // 16, 5 * h // 8      # 120x300 at 640x480
        frames[t, y0:y0 + bh, x0:x0 + bw] = np.clip(rng.normal(180, 30, (bh, bw, 3)), 0, 255).astype(np.uint8)
        boxes.append([x0, y0, bw, bh])
    return frames, np.array(boxes, np.float64)


def oracle_topdown(sd, width, frames_bgr, bboxes, image_size, post, kernel, pairs=hrnet.COCO_FLIP_PAIRS, num_joints=17):
    model = onets.HRNetRef(sd, width, num_joints=num_joints)
    out = []
    for fr, bb in zip(frames_bgr, bboxes):
        if np.isnan(bb).any():
            out.append(np.zeros((num_joints, 3)))
            continue
        t, c, s, _ = opre.top_down_input(fr[:, :, ::-1], bb, image_size)      # wrapper's BGR->RGB, then mmpose's swap
        hm = model.forward(t[None])
        hmf = model.forward(np.ascontiguousarray(t[None, :, :, ::-1]))
        k, _ = odec.decode_topdown(hm, hmf, pairs, c[None], s[None], post_process=post, kernel=kernel)
        out.append(k[0])
    return out


@pytest.mark.parametrize("k,pairs", [(136, hrnet.HALPE_FLIP_PAIRS), (133, hrnet.WHOLEBODY_FLIP_PAIRS)])
def test_wide_heads_and_flip_pairs(ctx, k, pairs):
    """MMPoseHalpe (the method scripts/process_h36m.py uses, K = 136) and MMPoseWholebody (K = 133): same backbone, wider
    head, their own flip pairs (wrappers/mmpose.py:41-52) -- crop -> backbone x 2 -> flip merge -> DARK decode against the
    oracle chain"""
    assert len(pairs) == 61
    perm = hrnet.flip_perm(k, pairs)
    assert np.array_equal(perm[perm], np.arange(k)) and (perm != np.arange(k)).sum() == 122
    spec = hrnet.HRNetSpec(32, k, 128, 96)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=6)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=8)
    td = ops.TopDown(net, k, flip_perm=perm, post="unbiased", blur_kernel=17)
    frames, bboxes = synth_clip(np.random.default_rng(4), 2, 240, 320)
    kp, valid = td.run(frames, np.arange(2, dtype=np.int32), bboxes)
    ref = oracle_topdown(sd, 32, frames, bboxes, (96, 128), "unbiased", 17, pairs, k)
    assert kp.shape == (2, k, 3)
    for i in range(2):
        assert np.array_equal(kp[i][:, 2], ref[i][:, 2].astype(np.float32))
        assert np.abs(kp[i][:, :2] - ref[i][:, :2]).max() <= 1e-3


def test_topdown_stage_matches_oracle_cascade(ctx):
    spec = hrnet.HRNetSpec(32, 17, 128, 96)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=16)
    td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="unbiased", blur_kernel=17)
    rng = np.random.default_rng(0)
    frames, bboxes = synth_clip(rng, 6, 240, 320)
    bboxes[2] = np.nan
    kp, valid = td.run(frames, np.arange(6, dtype=np.int32), bboxes)
    ref = oracle_topdown(sd, 32, frames, bboxes, (96, 128), "unbiased", 17)
    assert valid.tolist() == [1, 1, 0, 1, 1, 1]
    assert not kp[2].any()
    for i in (0, 1, 3, 4, 5):
        assert np.array_equal(kp[i][:, 2], ref[i][:, 2].astype(np.float32))            # scores bit-exact
        assert np.abs(kp[i][:, :2] - ref[i][:, :2]).max() <= 1e-3                     # north_star tolerance, px


# ---- the drop-in boundary: tables + wrappers (configs[0]) --------------------------------------------------
def test_populate_topdown_and_lifting(ctx, tmp_path, monkeypatch):
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import djshim, pipeline as pl, video
    from posepipeline_amd.wrappers import mmpose as wmm
    djshim.reset()
    rng = np.random.default_rng(0)                                    # config index 0
    frames, boxes = synth_clip(rng, 32, 480, 640)
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames, fps=30.0)
    vkey = {"video_project": "test", "filename": "clip"}
    pl.Video().insert1({**vkey, "video": path, "start_time": datetime.datetime(2024, 1, 1)})
    pl.VideoInfo().populate(vkey)
    assert (pl.VideoInfo & vkey).fetch1("num_frames", "height", "width") == (32, 480, 640)
    # tracking results as the tracking stage would store them (one track, two dropouts)
    tracks = [[{"track_id": 1, "tlbr": np.r_[b[:2], b[:2] + b[2:]], "tlhw": b, "confidence": 0.9}] for b in boxes]
    for t in (10, 11, 12, 13, 14, 20):
        tracks[t] = []
    tkey = {**vkey, "tracking_method": 5}
    pl.TrackingBboxMethod().insert1(tkey)
    pl.TrackingBbox().insert1({**tkey, "tracks": tracks, "num_tracks": 1})
    pl.PersonBboxValid().insert1({**tkey, "video_subject_id": 0, "keep_tracks": [1]})
    pl.PersonBbox().populate(tkey)
    bbox, present = (pl.PersonBbox & tkey).fetch1("bbox", "present")
    assert present.sum() == 31 and np.isnan(bbox[12]).all()          # 5-frame gap: 2 bfilled + 2 ffilled, 1 left
    # 2D: W32 256x192 member of the family (BASELINE configs[0]) through the same wrapper code path
    monkeypatch.setitem(wmm._METHODS, "HRNet_W48_COCO", wmm._METHODS["HRNet_W32_COCO"])
    wmm._cache.clear()
    pkey = {**tkey, "video_subject_id": 0, "top_down_method": 0}
    pl.TopDownMethod().insert1(pkey)
    pl.TopDownPerson().populate(pkey)
    kp = (pl.TopDownPerson & pkey).fetch1("keypoints")
    assert kp.shape == (32, 17, 3) and kp.dtype == np.float64       # zero rows are float64 -> stacked float64
    assert not kp[12].any() and kp[0].any()
    spec = hrnet.hrnet_w32_256x192()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    for i in (0, 11, 31):
        ref = oracle_topdown(sd, 32, frames[i:i + 1], bbox[i:i + 1], (192, 256), "default", 11)[0]
        assert np.abs(kp[i][:, :2] - ref[:, :2]).max() <= 1e-3
        assert np.array_equal(kp[i][:, 2].astype(np.float32), ref[:, 2].astype(np.float32))
    # 3D lifting on the stored 2D track
    lkey = {**pkey, "lifting_method": 1}
    pl.LiftingMethod().insert1(lkey)
    pl.LiftingPerson().populate(lkey)
    k3d, kvalid = (pl.LiftingPerson & lkey).fetch1("keypoints_3d", "keypoints_valid")
    assert k3d.shape == (32, 17, 3) and k3d.dtype == np.float64 and kvalid == [True] * 32
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    vspec = vp3d.VideoPose3DSpec()
    vsd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vspec), seed=3)
    kn = normalize_screen_coordinates(kp[:, :, :2], 640, 480).astype("float32")
    ref3d = onets.VideoPose3DRef(vsd).forward(onets.videopose3d_windows(kn, 121))
    assert np.array_equal(k3d.astype(np.float32), ref3d)
    wmm._cache.clear()
