"""ViTPose top-down stage (UDP crop -> ViT program -> flip-merge + DARK-UDP decode) vs the CPU oracle.

The UDP crop is integer / lookup arithmetic: bit-exact.  The decode is float arithmetic on given heatmaps: 1e-3 px
(north_star tolerance).  The fused stage is checked by composition: its input tensor and its keypoints are compared
with the oracle applied to the SAME frames / the SAME (GPU) heatmaps; the bf16 network itself is covered by
tests/test_gpu_vit.py with its own tolerance, because argmax over heatmaps of a randomly initialised network is not
stable under 1e-3 perturbations.
"""
import numpy as np
import pytest

from oracle import decode as odec
from oracle import preprocess as opre
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet
from posepipeline_amd.models import vitpose as MV
from posepipeline_amd.program import Net
from tests.test_gpu_stages import synth_frames

pytestmark = pytest.mark.gpu

BBOXES = np.array([
    [100.3, 40.7, 80.2, 190.9],
    [-30.5, -20.25, 200.0, 150.0],     # over the top-left corner
    [400.0, 200.0, 120.0, 100.0],      # over the bottom-right corner, wide box
    [np.nan, np.nan, np.nan, np.nan],  # absent person
    [10.0, 10.0, 30.0, 60.0],
    [0.0, 0.0, 480.0, 270.0],
], dtype=np.float64)
FIDX = np.array([0, 1, 2, 0, 1, 2], dtype=np.int32)


def test_udp_crop_bit_exact(ctx):
    rng = np.random.default_rng(21)
    frames = synth_frames(rng, 3, 270, 480)
    res = ops.crop_affine_normalize(ctx, frames, FIDX, BBOXES, out_wh=(192, 256), flip=True, want_crop_u8=True, udp=True)
    out = res["out"]
    for i, bb in enumerate(BBOXES):
        if np.isnan(bb).any():
            assert res["valid"][i] == 0 and not out[i].any()
            continue
        t, c, s, crop = opre.top_down_input_udp(frames[FIDX[i]][:, :, ::-1], bb, (192, 256))
        assert np.array_equal(res["center_scale"][i], np.concatenate([c, s]))
        assert np.array_equal(res["crop_u8"][i], crop), f"person {i}: {np.abs(res['crop_u8'][i].astype(int) - crop).max()}"
        assert np.array_equal(np.transpose(out[i][:, :, :3], (2, 0, 1)), t)
        assert np.array_equal(out[len(BBOXES) + i], out[i][:, ::-1])
    # the UDP transform is not the 3-point one: a half-pixel-order shift must be visible against the plain crop
    plain = ops.crop_affine_normalize(ctx, frames, FIDX, BBOXES, out_wh=(192, 256), flip=False, want_crop_u8=True)
    assert not np.array_equal(plain["crop_u8"][0], res["crop_u8"][0])


def _gaussian_maps(rng, n, k, h, w, perm):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.zeros((n, k, h, w), np.float32)
    hmf = np.zeros_like(hm)
    for i in range(n):
        for j in range(k):
            cx, cy = rng.uniform(0, w - 1), rng.uniform(0, h - 1)
            if j == 3:
                cx, cy = 0.2, 0.4              # corner: edge-replicated stencil
            if j == 4:
                cx, cy = w - 1.3, h - 1.2
            g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2)).astype(np.float32)
            hm[i, j] = g * rng.uniform(0.3, 1.0) + rng.normal(0, 0.005, (h, w))
            gf = np.exp(-((xx - (w - 1 - cx)) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2)).astype(np.float32)
            hmf[i, perm[j]] = gf * rng.uniform(0.3, 1.0) + rng.normal(0, 0.005, (h, w))
    return hm, hmf


def test_dark_udp_decode(ctx):
    rng = np.random.default_rng(8)
    n, k, h, w = 3, 17, 64, 48
    perm = hrnet.flip_perm(k)
    hm, hmf = _gaussian_maps(rng, n, k, h, w, perm)
    hm[1, 7] = -np.abs(hm[1, 7]) - 1e-3         # no positive peak: coordinates stay at -1 (stated deviation)
    hmf[1, perm[7]] = -np.abs(hmf[1, perm[7]]) - 1e-3
    center = rng.uniform(100, 900, (n, 2)).astype(np.float32)
    scale = rng.uniform(0.5, 4.0, (n, 2)).astype(np.float32)
    ref, ref_merged = odec.decode_topdown_udp(hm, hmf, hrnet.COCO_FLIP_PAIRS, center, scale, kernel=11)
    got, merged = ops.flip_merge_decode(ctx, hm, hmf, np.concatenate([center, scale], 1), flip_perm=perm, shift_heatmap=False,
                                        post="udp", blur_kernel=11, want_merged=True)
    assert np.array_equal(merged, ref_merged)
    assert np.array_equal(got[:, :, 2], ref[:, :, 2])
    err = np.abs(got[:, :, :2] - ref[:, :, :2]).max()
    assert err <= 1e-3, err
    # sub-pixel refinement actually happened: decoded != plain argmax back-mapping
    plain, _ = ops.flip_merge_decode(ctx, hm, hmf, np.concatenate([center, scale], 1), flip_perm=perm, shift_heatmap=False,
                                     post=None)
    assert np.abs(got[:, :, :2] - plain[:, :, :2]).max() > 0.05


def test_vitpose_method_through_the_wrapper(ctx, tmp_path, monkeypatch):
    """`mmpose_top_down_person(key, method="ViTPose_B_COCO")` on the table shim: same table reads, same return contract
    (zero rows for absent frames -> float64 stack) as the HRNet methods; a 2-block encoder keeps it quick."""
    import datetime
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import djshim, pipeline as pl, video
    from posepipeline_amd.wrappers import mmpose as wmm
    from tests.test_gpu_pipeline import synth_clip
    djshim.reset()
    rng = np.random.default_rng(2)
    frames, boxes = synth_clip(rng, 10, 240, 320)
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames, fps=30.0)
    vkey = {"video_project": "test", "filename": "clip"}
    pl.Video().insert1({**vkey, "video": path, "start_time": datetime.datetime(2024, 1, 1)})
    tracks = [[{"track_id": 1, "tlbr": np.r_[b[:2], b[:2] + b[2:]], "tlhw": b, "confidence": 0.9}] for b in boxes]
    for t in (3, 4, 5, 6, 7):
        tracks[t] = []
    tkey = {**vkey, "tracking_method": 5}
    pl.TrackingBboxMethod().insert1(tkey)
    pl.TrackingBbox().insert1({**tkey, "tracks": tracks, "num_tracks": 1})
    pl.PersonBboxValid().insert1({**tkey, "video_subject_id": 0, "keep_tracks": [1]})
    pl.PersonBbox().populate(tkey)
    bbox = (pl.PersonBbox & tkey).fetch1("bbox")
    small = lambda k: MV.VitPoseSpec(dim=768, depth=2, heads=12, num_joints=k, deconv=(64, 64))   # noqa: E731
    monkeypatch.setitem(wmm._METHODS, "ViTPose_B_COCO", (small,) + wmm._METHODS["ViTPose_B_COCO"][1:])
    wmm._cache.clear()
    pkey = {**tkey, "video_subject_id": 0}
    kp = wmm.mmpose_top_down_person(pkey, method="ViTPose_B_COCO")
    assert kp.shape == (10, 17, 3) and kp.dtype == np.float64
    absent = np.isnan(bbox).any(axis=1)
    assert absent.sum() == 1 and not kp[absent].any() and all(kp[i].any() for i in np.flatnonzero(~absent))
    # the stage really ran the UDP decode on this model's maps: recompute from the program's own buffers
    _, net, td, _ = wmm._cache[("ViTPose_B_COCO", 0)]
    n = 10
    hm = net.read("output", 2 * n).reshape(2 * n, 17, 64, 48)
    cs = np.stack([np.concatenate(opre.box2cs(b, (192, 256))) if not np.isnan(b).any() else np.zeros(4, np.float32) for b in bbox])
    ref, _ = odec.decode_topdown_udp(hm[:n], hm[n:], hrnet.COCO_FLIP_PAIRS, cs[:, :2], cs[:, 2:], kernel=11)
    ok = ~absent
    assert np.abs(kp[ok][:, :, :2] - ref[ok][:, :, :2]).max() <= 1e-3
    with pytest.raises(UnboundLocalError):
        wmm.mmpose_top_down_person(pkey, method="ViTPose_XXL")
    # the same through the table layer: lookup row 100 ("ViTPoseB", an extension of TopDownMethodLookup) -> TopDownPerson
    tdkey = {**pkey, "top_down_method": 100}
    assert (pl.TopDownMethodLookup & {"top_down_method": 100}).fetch1("top_down_method_name") == "ViTPoseB"
    pl.TopDownMethod().insert1(tdkey)
    pl.TopDownPerson().populate(tdkey)
    assert np.array_equal((pl.TopDownPerson & tdkey).fetch1("keypoints"), kp)
    wmm._cache.clear()


def test_cascade_with_vitpose_2d_stage(ctx):
    """bench.py --workload cascade5 in miniature: detector -> tracker -> ViTPose (UDP) -> VideoPose3D; the 2D stage of
    the cascade equals the stand-alone top-down stage on the same frames and boxes, the 3D stage the oracle on those."""
    from oracle import nets as onets
    from posepipeline_amd.cascade import Cascade
    from posepipeline_amd.models import faster_rcnn as fr, synth
    from posepipeline_amd.models import videopose3d as vp3d
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    from tests.test_gpu_detector import synth_frame
    rng = np.random.default_rng(13)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(4)])
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    spec = MV.VitPoseSpec(dim=640, depth=2, heads=8, deconv=(64, 64))
    p = MV.synth_params(spec, seed=6)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    cas = Cascade(ctx, det_sd, p, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=spec)
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(4)]
    out = [cas.step(frames[0:2], replay=gt[0:2]), cas.step(frames[2:4], replay=gt[2:4]), cas.flush()]
    tid = out[0]["tracks"][0][0][0]
    k2 = np.concatenate([o["keypoints"][tid] for o in out[:2]])
    assert k2.shape == (4, 17, 3)
    boxes = np.array([[g[0, 0], g[0, 1], g[0, 2] - g[0, 0], g[0, 3] - g[0, 1]] for g in gt], np.float64)
    net = Net(ctx, MV.build_vitpose_program(spec, p), max_batch=8)
    td = ops.TopDown(net, num_joints=17, flip_perm=hrnet.flip_perm(17), shift_heatmap=False, post="udp", blur_kernel=11)
    ref, _ = td.run(frames, np.arange(4, dtype=np.int32), boxes)
    # same kernels, same inputs, other batch size (2 + 2 vs 4): the tile schedule does not depend on the batch, so equal
    assert np.array_equal(k2, ref)
    kn = normalize_screen_coordinates(k2[:, :, :2], w, h).astype(np.float32)      # float32 track: the reference's float32 path
    ref3 = onets.VideoPose3DRef(lift_sd).forward(onets.videopose3d_windows(kn, 121))
    assert np.array_equal(out[2]["keypoints_3d"][tid], ref3)                     # whole clip, emitted at flush
    td.close()
    net.close()


def test_vitpose_topdown_composition(ctx):
    spec = MV.VitPoseSpec(dim=640, depth=2, heads=8, mlp_ratio=4, num_joints=17, deconv=(64, 64))
    p = MV.synth_params(spec, seed=4)
    n = len(BBOXES)
    net = Net(ctx, MV.build_vitpose_program(spec, p), max_batch=2 * n)
    perm = hrnet.flip_perm(17)
    td = ops.TopDown(net, num_joints=17, flip_perm=perm, shift_heatmap=False, post="udp", blur_kernel=11)
    rng = np.random.default_rng(31)
    frames = synth_frames(rng, 3, 270, 480)
    kp, valid = td.run(frames, FIDX, BBOXES)
    assert list(valid) == [1, 1, 1, 0, 1, 1]
    x = net.read("input", 2 * n)
    hm = net.read("output", 2 * n).reshape(2 * n, 17, *spec.heatmap_hw)
    cs = []
    for i, bb in enumerate(BBOXES):
        if np.isnan(bb).any():
            assert not x[i].any() and not kp[i].any()
            cs.append(np.zeros(4, np.float32))
            continue
        t, c, s, _ = opre.top_down_input_udp(frames[FIDX[i]][:, :, ::-1], bb, (192, 256))
        assert np.array_equal(np.transpose(x[i][:, :, :3], (2, 0, 1)), t)
        assert np.array_equal(x[n + i], x[i][:, ::-1])
        cs.append(np.concatenate([c, s]))
    cs = np.stack(cs)
    ref, _ = odec.decode_topdown_udp(hm[:n], hm[n:], hrnet.COCO_FLIP_PAIRS, cs[:, :2], cs[:, 2:], kernel=11)
    ok = valid.astype(bool)
    assert np.array_equal(kp[ok][:, :, 2], ref[ok][:, :, 2])
    err = np.abs(kp[ok][:, :, :2] - ref[ok][:, :, :2]).max()
    assert err <= 1e-3, err
    td.close()
    net.close()
