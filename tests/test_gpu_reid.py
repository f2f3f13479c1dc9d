"""ReID branch of mmtrack's DeepSORT configuration on the GPU against oracle/reid_mm.py: the crop kernel
(F.interpolate bilinear of integer rectangles of the detector's input tensor), the ResNet-50 ReID program (AvgPool neck,
Linear + BN1d + ReLU, fc_out) and the wrapper end to end -- all bit-exact (same float32 operations in the same order)."""
import ctypes as C

import numpy as np
import pytest

from oracle import detector as odet
from oracle import reid_mm as orm
from posepipeline_amd import _lib as L
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import reid_r50, synth
from posepipeline_amd.program import Net
from tests.test_gpu_detector import synth_frame

pytestmark = pytest.mark.gpu


def test_crop_resize_bilinear_bit_exact(ctx):
    rng = np.random.default_rng(2)
    src = rng.standard_normal((2, 40, 56, 4)).astype(np.float32)
    rects = np.array([[0, 0, 0, 56, 40], [1, 5, 7, 6, 39], [0, 10, 3, 31, 4], [1, 20, 10, 50, 38], [0, 55, 39, 56, 40]], np.int32)
    d_src, d_out = ctx.malloc(src.nbytes), ctx.malloc(len(rects) * 256 * 128 * 16)
    ctx.h2d(d_src, src)
    L.check(ctx.lib.pp_crop_resize_bilinear(ctx.handle, C.c_void_p(d_src), 2, 40, 56, L.ptr(rects), len(rects), 256, 128, C.c_void_p(d_out)),
            "pp_crop_resize_bilinear")
    got = np.empty((len(rects), 256, 128, 4), np.float32)
    ctx.d2h(got, d_out)
    for r, (f, x1, y1, x2, y2) in enumerate(rects):
        assert np.array_equal(got[r], orm.interpolate_bilinear(src[f, y1:y2, x1:x2], (256, 128))), r
    bad = np.array([[0, 0, 0, 57, 40]], np.int32)
    with pytest.raises(L.PosePipeHipError, match="outside"):
        L.check(ctx.lib.pp_crop_resize_bilinear(ctx.handle, C.c_void_p(d_src), 2, 40, 56, L.ptr(bad), 1, 256, 128, C.c_void_p(d_out)), "x")
    ctx.free(d_src)
    ctx.free(d_out)


def test_reid_network_bit_exact(ctx):
    sd = synth.synth_state_dict(reid_r50.reid_param_shapes(), seed=7)
    prog = reid_r50.build_reid_program(sd)
    net = Net(ctx, prog, max_batch=3)
    x = np.zeros((3, 256, 128, 4), np.float32)
    x[..., :3] = np.random.default_rng(1).standard_normal((3, 256, 128, 3)).astype(np.float32)
    got = net.forward(x, out_name="features").reshape(3, 128)
    ref = orm.ReidNetRef(sd).forward(x)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-4
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    net.close()


def test_mmtrack_deepsort_wrapper_with_reid_matches_oracle(ctx, tmp_path, monkeypatch):
    """mmtrack_bounding_boxes(path, "deepsort") = detector -> kept detections -> ReID crops of the detector input -> embeddings
    -> SortTracker with ReID, against the oracle chain frame by frame: ids bit-exact, boxes and scores equal"""
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import video
    from posepipeline_amd.wrappers import mmtrack as wmt
    wmt._cache.clear()
    rng = np.random.default_rng(2)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(4)])
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, frames)
    tracks = wmt.mmtrack_bounding_boxes(path, "deepsort")
    assert len(tracks) == 4
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    reid_sd = synth.synth_state_dict(reid_r50.reid_param_shapes(), seed=7)
    model, reid, trk = odet.FasterRCNNRef(det_sd), orm.ReidNetRef(reid_sd), orm.SortReidTrackerRef()
    n_total = 0
    for f in range(4):
        rgb = frames[f][:, :, ::-1]                                              # wrapper: BGR -> RGB, then mmtrack
        dets = odet.detect(model, rgb)
        inp, sf, (nh, nw) = odet.preprocess(rgb)
        dets = dets[trk.keep(dets)]
        emb = reid.forward(orm.crop_imgs(inp, dets[:, :4], sf, (nh, nw))) if len(dets) else np.zeros((0, 128), np.float32)
        rows = trk.step(dets, emb, f)
        assert len(tracks[f]) == len(rows), f
        for d, x in zip(tracks[f], rows):
            assert isinstance(d["track_id"], int) and d["track_id"] == int(x[0])
            assert np.array_equal(d["tlbr"], x[1:5]) and d["confidence"] == x[5]
        n_total += len(rows)
    assert n_total > 0
    # without the appearance branch the same call is the sort_faster-rcnn configuration
    monkeypatch.setenv("POSEPIPE_MMTRACK_REID", "0")
    wmt._cache.clear()
    from oracle.tracking import SortTrackerRef
    tracks0 = wmt.mmtrack_bounding_boxes(path, "deepsort")
    ref_trk = SortTrackerRef()
    for f in range(4):
        rows = ref_trk.step(odet.detect(model, frames[f][:, :, ::-1]))
        assert [d["track_id"] for d in tracks0[f]] == [int(x[0]) for x in rows]
    wmt._cache.clear()


def test_cascade_with_reid_tracker_keeps_identity_through_a_gap(ctx):
    """Cascade(reid_sd=...): the DeepSORT configuration inside the streamed cascade.  A person missed for five frames keeps
    its id (appearance match; SORT alone would issue a new one), so PersonStreams sees ONE track with a gap: 2 frames
    forward-filled, 1 zero row, 2 back-filled."""
    from posepipeline_amd.cascade import Cascade, collect
    from posepipeline_amd.models import hrnet
    from posepipeline_amd.models import videopose3d as vp3d
    rng = np.random.default_rng(4)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(10)])
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    reid_sd = synth.synth_state_dict(reid_r50.reid_param_shapes(), seed=7)
    spec = hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=5, max_persons=2, pose_spec=spec, reid_sd=reid_sd)
    gt = [np.array([[60 + 2 * t, 20, 130 + 2 * t, 120, 0.9]], np.float32) for t in range(10)]
    for t in (3, 4, 5, 6, 7):
        gt[t] = np.zeros((0, 5), np.float32)
    outs = [cas.step(frames[0:5], replay=gt[0:5]), cas.step(frames[5:10], replay=gt[5:10]), cas.flush()]
    ids = [[r[0] for r in fr_] for o in outs for fr_ in o["tracks"]]
    assert ids == [[0], [0], [0], [], [], [], [], [], [0], [0]]
    f2, k2 = collect(outs, "keypoints")[0]
    assert f2 == 0 and len(k2) == 10
    # bfill(2) fills frames 6, 7 from frame 8, ffill(2) frames 3, 4 from frame 2, frame 5 stays a zero row
    assert [bool(r.any()) for r in k2] == [True] * 5 + [False] + [True] * 4
