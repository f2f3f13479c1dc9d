"""End-to-end GPU checks of the tracking wrapper and of the chunked cascade against the CPU oracle chain."""
import numpy as np
import pytest

from oracle import detector as odet
from oracle import nets as onets
from oracle.tracking import SortTrackerRef
from posepipeline_amd.models import faster_rcnn as fr
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.models import videopose3d as vp3d
from tests.test_gpu_detector import synth_frame
from tests.test_gpu_pipeline import oracle_topdown

pytestmark = pytest.mark.gpu


def test_mmtrack_wrapper_matches_oracle(ctx, tmp_path, monkeypatch):
    """the sort_faster-rcnn configuration (appearance branch off); the ReID branch: tests/test_gpu_reid.py"""
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("POSEPIPE_MMTRACK_REID", "0")
    from posepipeline_amd import video
    from posepipeline_amd.wrappers import mmtrack as wmt
    wmt._cache.clear()
    rng = np.random.default_rng(2)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(3)])
    path = str(tmp_path / "v.ppvid")
    video.write_ppvid(path, frames)
    with pytest.raises(Exception, match="Unknown config file"):
        wmt.mmtrack_bounding_boxes(path, "nope")
    tracks = wmt.mmtrack_bounding_boxes(path, "deepsort")
    assert len(tracks) == 3
    sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    model = odet.FasterRCNNRef(sd)
    ref_trk = SortTrackerRef()
    for f in range(3):
        rows = ref_trk.step(odet.detect(model, frames[f][:, :, ::-1]))      # wrapper: BGR->RGB, then mmtrack
        assert len(tracks[f]) == len(rows)
        for d, x in zip(tracks[f], rows):
            assert isinstance(d["track_id"], int) and d["track_id"] == int(x[0])          # ids bit-exact
            assert np.array_equal(d["tlbr"], x[1:5])
            assert np.array_equal(d["tlhw"], np.array([x[1], x[2], x[3] - x[1], x[4] - x[2]]))
            assert d["confidence"] == x[5]
    wmt._cache.clear()


def _setup(h, w, pose_spec=None):
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_spec = pose_spec or hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    return det_sd, pose_spec, pose_sd, lift_sd


def reference_3d(k2_rows, first, n_frames, w, h, lift_sd):
    """LiftingPerson for one subject the way the reference computes it (wrappers/videopose3d.py:23-91) from the stored 2D
    track: rows outside [first, first + len) are the zeros((K, 3)) of absent frames (wrappers/mmpose.py:67-69), which make
    the stacked array float64; one edge-replicated 243-frame window per frame of the WHOLE clip."""
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    rows = [np.zeros((17, 3))] * first + [r[:17] for r in k2_rows] + [np.zeros((17, 3))] * (n_frames - first - len(k2_rows))
    rows = [np.zeros((17, 3)) if not r.any() else r for r in rows]           # decided-absent rows are float64 zeros, too
    kp = np.asarray(rows)
    kn = normalize_screen_coordinates(kp[:, :, :2], w, h).astype("float32")
    return onets.VideoPose3DRef(lift_sd).forward(onets.videopose3d_windows(kn, 121))


def test_cascade_chunks_match_oracle_chain(ctx):
    from posepipeline_amd.cascade import Cascade, collect
    rng = np.random.default_rng(3)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(4)])
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    # replayed person boxes (random-weight detector): one person drifting right
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(4)]
    out = [cas.step(frames[0:2], replay=gt[0:2]), cas.step(frames[2:4], replay=gt[2:4]), cas.flush()]
    assert [len(t) for o in out for t in o["tracks"]] == [1, 1, 1, 1]
    tid = out[0]["tracks"][0][0][0]
    assert all(t[0][0] == tid for o in out for t in o["tracks"])                      # one identity throughout
    assert [o["keypoints_frames"][tid].tolist() for o in out[:2]] == [[0, 1], [2, 3]] and not out[2]["keypoints"]
    k2 = np.concatenate([o["keypoints"][tid] for o in out[:2]])
    boxes = np.array([[g[0, 0], g[0, 1], g[0, 2] - g[0, 0], g[0, 3] - g[0, 1]] for g in gt], np.float64)
    ref2 = oracle_topdown(pose_sd, 32, frames, boxes, (96, 128), "unbiased", 17)
    for i in range(4):
        assert np.abs(k2[i][:, :2] - ref2[i][:, :2]).max() <= 1e-3
        assert np.array_equal(k2[i][:, 2], ref2[i][:, 2].astype(np.float32))
    # a frame is lifted when its +121 look-ahead exists: nothing before the end of this 4-frame clip, everything at flush
    assert not out[0]["keypoints_3d"] and not out[1]["keypoints_3d"]
    assert out[2]["keypoints_3d_frames"][tid].tolist() == [0, 1, 2, 3]
    assert np.array_equal(out[2]["keypoints_3d"][tid], reference_3d(k2, 0, 4, w, h, lift_sd))
    assert collect(out)[tid][0] == 0
    # and the detector really ran on the chunk: its own boxes equal the oracle's
    d0 = cas.detector.run(frames[:1])[0]
    assert np.array_equal(d0, odet.detect(odet.FasterRCNNRef(det_sd), frames[0][:, :, ::-1]))


def test_cascade_long_clip_every_chunk_equals_whole_clip_lifting(ctx):
    """N = 330 frames in 6 chunks: the 3D joints EVERY step emits equal process_videopose3d's whole-clip result
    (window [t-121, t+121], edge replication at the two ends of the clip only) bit for bit; frame t leaves the cascade with
    the chunk that brings frame t+121."""
    from posepipeline_amd.cascade import Cascade, collect
    rng = np.random.default_rng(8)
    h, w, n, chunk = 135, 240, 330, 64
    base = np.stack([synth_frame(rng, h, w) for _ in range(6)])
    frames = base[np.arange(n) % 6].copy()
    for t in range(n):                                   # make every frame different where the person is
        frames[t, 30:100, 20 + t // 3: 60 + t // 3] = (frames[t, 30:100, 20 + t // 3: 60 + t // 3].astype(np.int32) + 3 * t) % 256
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=1, pose_spec=pose_spec)
    gt = [np.array([[20 + 0.4 * t, 20, 90 + 0.4 * t, 120, 0.9]], np.float32) for t in range(n)]
    outs = [cas.step(frames[i:i + chunk], replay=gt[i:i + chunk]) for i in range(0, n, chunk)] + [cas.flush()]
    ids = {r[0] for o in outs for fr_ in o["tracks"] for r in fr_}
    assert len(ids) == 1
    tid = ids.pop()
    f2, k2 = collect(outs, "keypoints")[tid]
    assert f2 == 0 and k2.shape == (n, 17, 3)
    # 2D against the oracle on sampled frames (first / last of chunks, the clip ends)
    boxes = np.array([[g[0, 0], g[0, 1], g[0, 2] - g[0, 0], g[0, 3] - g[0, 1]] for g in gt], np.float64)
    for i in (0, 63, 64, 200, n - 1):
        ref = oracle_topdown(pose_sd, 32, frames[i:i + 1], boxes[i:i + 1], (96, 128), "unbiased", 17)[0]
        assert np.abs(k2[i][:, :2] - ref[:, :2]).max() <= 1e-3 and np.array_equal(k2[i][:, 2], ref[:, 2].astype(np.float32))
    ref3 = reference_3d(k2, 0, n, w, h, lift_sd)
    emitted = []
    for k, o in enumerate(outs):
        if tid not in o["keypoints_3d"]:
            emitted.append(0)
            continue
        fr3 = o["keypoints_3d_frames"][tid]
        emitted.append(len(fr3))
        assert np.array_equal(o["keypoints_3d"][tid], ref3[fr3]), (k, np.abs(o["keypoints_3d"][tid] - ref3[fr3]).max())
    # chunk k (frames [64k, 64k+64)) releases the frames up to 64k + 63 - 121
    assert emitted == [0, 7, 64, 64, 64, 10, 121]
    assert not cas.persons.streams and max(len(o["keypoints_3d"]) for o in outs) == 1


def test_cascade_1080p_four_persons_matches_table_chain(ctx):
    """BASELINE.json configs[2]: multi-person 1080p through detector -> SORT -> HRNet-W48 384x288 (-> lifting), 4 persons
    with crossing trajectories and one missed detection at a chunk boundary.  Track ids bit-exact against the oracle
    tracker; for EVERY id, boxes = PersonBbox.make(keep_tracks=[id]) (back-fills reach into the previous chunk: crops from
    the device tail buffer), 2D within 1e-3 px of the oracle chain on the sampled person-frames, 3D bit-equal to the
    whole-clip lifting of the stored 2D track."""
    from posepipeline_amd.cascade import Cascade, collect
    from posepipeline_amd.tracking import person_bbox
    from posepipeline_amd.video import ArrayVideo
    rng = np.random.default_rng(21)
    h, w, n, chunk = 1080, 1920, 16, 8
    bg = rng.integers(40, 200, (h // 40, w // 40, 3)).astype(np.uint8)
    bg = np.repeat(np.repeat(bg, 40, axis=0), 40, axis=1)
    people = [dict(x=200.0, y=150.0, w=170, h=520, vx=38.0), dict(x=900.0, y=260.0, w=150, h=470, vx=-36.0),
              dict(x=1400.0, y=90.0, w=190, h=580, vx=6.0), dict(x=100.0, y=500.0, w=120, h=330, vx=12.0)]
    for q in people:
        q["tex"] = rng.integers(0, 256, (q["h"], q["w"], 3)).astype(np.uint8)
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
    det_sd, _, pose_sd, lift_sd = _setup(h, w, spec)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=4, pose_spec=spec)
    outs = list(cas.run_video(ArrayVideo(frames), replay_fn=lambda first, m: gt[first:first + m]))
    assert [o["first_frame"] for o in outs] == [0, 8, 16]
    tracks = [fr_ for o in outs for fr_ in o["tracks"]]
    # ids: bit-exact against the oracle tracker on the same detections
    ref_trk = SortTrackerRef()
    for t in range(n):
        rows = ref_trk.step(gt[t])
        assert [r[0] for r in tracks[t]] == [int(x[0]) for x in rows]
        assert np.array_equal(np.array([r[1:] for r in tracks[t]], np.float32), rows[:, 1:])
    ids = sorted({r[0] for fr_ in tracks for r in fr_})
    assert len(ids) == 5                                                    # the missed person comes back under a new id
    k2, k3 = collect(outs, "keypoints"), collect(outs, "keypoints_3d")
    assert sorted(k2) == sorted(k3) == ids
    dicts = [[{"track_id": r[0], "tlhw": np.array([r[1], r[2], r[3] - r[1], r[4] - r[2]], np.float64)} for r in fr_] for fr_ in tracks]
    sample = (0, 7, 8, 9, 15)
    for tid in ids:
        bbox, present = person_bbox(dicts, [tid])
        f2, a2 = k2[tid]
        filled = np.flatnonzero(present)
        assert f2 == filled[0] and f2 + len(a2) > filled[-1]
        for t in range(f2, f2 + len(a2)):                                   # zero rows exactly where the reference has none
            assert a2[t - f2].any() == bool(present[t]), (tid, t)
        for t in sample:
            if not present[t]:
                continue
            ref = oracle_topdown(pose_sd, 48, frames[t:t + 1], bbox[t:t + 1], (288, 384), "unbiased", 17)[0]
            assert np.abs(a2[t - f2][:, :2] - ref[:, :2]).max() <= 1e-3, (tid, t)
            assert np.array_equal(a2[t - f2][:, 2], ref[:, 2].astype(np.float32)), (tid, t)
        f3, a3 = k3[tid]
        ref3 = reference_3d(a2, f2, n, w, h, lift_sd)
        assert f3 == f2 and np.array_equal(a3, ref3[f3:f3 + len(a3)]), tid
    # the two fills around the miss: the old id is forward-filled over frames 8 and 9, the new id (born in frame 9)
    # back-filled over frames 7 and 8 -- frame 7 belongs to the previous chunk
    old = tracks[0][2][0]
    new = [r[0] for r in tracks[9] if r[0] not in {q[0] for q in tracks[0]}][0]
    assert k2[new][0] == 7 and k2[old][1][8].any() and k2[old][1][9].any() and not k2[old][1][10:].any()


def test_streamed_video_equals_host_chunks(ctx, tmp_path):
    """run_video (reader thread -> page-locked staging -> copy stream, ragged last chunk, device tail buffer) returns
    exactly what step() / flush() return on the same frames passed from host memory (a self-consistency property of the
    two input modes, not a parity test; parity of either is checked above)."""
    from posepipeline_amd import video
    from posepipeline_amd.cascade import Cascade
    rng = np.random.default_rng(5)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(5)])
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames)
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(5)]
    gt[2] = np.zeros((0, 5), np.float32)                 # a miss: new id in frame 3, back-filled over frames 1 and 2
    ref = [cas.step(frames[i:i + 2], replay=gt[i:i + 2]) for i in (0, 2, 4)] + [cas.flush()]
    cas.reset()
    got = list(cas.run_video(video.open_video(path), replay_fn=lambda first, n: gt[first:first + n]))
    assert [o["first_frame"] for o in got] == [0, 2, 4, 5]
    assert sorted(ref[1]["keypoints_frames"]) == [0, 1] and ref[1]["keypoints_frames"][1].tolist() == [1, 2, 3]
    for a, b in zip(ref, got):
        assert a["tracks"] == b["tracks"]
        for what in ("keypoints", "keypoints_3d"):
            assert a[what].keys() == b[what].keys()
            for tid in a[what]:
                assert np.array_equal(a[what][tid], b[what][tid])
                assert np.array_equal(a[what + "_frames"][tid], b[what + "_frames"][tid])
    # the detector consumes the streamed frames (no replay): same tracks as from host memory
    rev = frames[::-1].copy()
    cas.reset()
    ref_first = cas.step(rev[:2])
    cas.reset()
    gen = cas.run_video(video.open_video(rev))
    first = next(gen)
    gen.close()
    assert first["tracks"] == ref_first["tracks"]


def test_cascade_with_default_tracking_method(ctx):
    """tracking="DeepSortYOLOv4" (tracking_method 0): YOLOv4 + mars-small128 + DeepSORT feed the same 2D / 3D stages;
    the 2D stage crops the tracker's (Kalman) boxes, like PersonBbox does with the reference's track dicts."""
    from posepipeline_amd.cascade import Cascade
    from posepipeline_amd.models import mars, yolov4
    rng = np.random.default_rng(11)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(4)])
    ysd = yolov4.synth_params(yolov4.yolov4_param_shapes(), seed=4, head_bias=-2.0)
    msd = yolov4.synth_params(mars.mars_param_shapes(), seed=5)
    pose_spec = hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    cas = Cascade(ctx, (ysd, msd), pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec,
                  tracking="DeepSortYOLOv4")
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(4)]
    out = [cas.step(frames[0:2], replay=gt[0:2]), cas.step(frames[2:4], replay=gt[2:4]), cas.flush()]
    tracks = [t for o in out for t in o["tracks"]]
    assert [len(t) for t in tracks] == [1, 1, 1, 1] and len({t[0][0] for t in tracks}) == 1     # one identity (id 1)
    assert tracks[0][0][0] == 1                                                                 # deep_sort ids start at 1
    tid = tracks[0][0][0]
    k2 = np.concatenate([o["keypoints"][tid] for o in out[:2]])
    boxes = np.array([t[0][6] for t in tracks], np.float64)              # the tracker's to_tlwh(), as parser.py:80 stores it
    # first frame: the Kalman mean is initialised from the detection, so the track box is the (int-truncated) input box
    assert np.allclose(boxes[0], [60, 20, 70, 100])
    ref2 = oracle_topdown(pose_sd, 32, frames, boxes, (96, 128), "unbiased", 17)
    for i in range(4):
        assert np.abs(k2[i][:, :2] - ref2[i][:, :2]).max() <= 1e-3
    assert np.array_equal(out[2]["keypoints_3d"][tid], reference_3d(k2, 0, 4, w, h, lift_sd))


def test_cascade_steady_state_keeps_device_memory_flat(ctx):
    """A long-running cascade (tracks born and lost all the time: every 96 frames the person jumps to another place, so a new id
    and a new person stream replace the old ones) must not grow: device memory in use after 12 chunks equals the level after 4
    (hipMemGetInfo of the runtime the library itself is linked to; one allocation granule of slack), and no person stream outlives its track's lifting tail."""
    import ctypes

    from posepipeline_amd.cascade import Cascade
    hip = ctypes.CDLL("libamdhip64.so")

    def device_bytes_in_use():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return total.value - free.value
    rng = np.random.default_rng(5)
    h, w, chunk, steps = 135, 240, 32, 12
    base = np.stack([synth_frame(rng, h, w) for _ in range(4)])
    det_sd, pose_spec, pose_sd, lift_sd = _setup(h, w)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=chunk, max_persons=2, pose_spec=pose_spec)

    def boxes(t):
        x = 15.0 + 90.0 * ((t // 96) % 2)                 # two places 90 px apart: no IoU between them -> a new track
        return np.array([[x, 20, x + 60, 120, 0.9]], np.float32)

    used, ids, live = [], set(), []
    for k in range(steps):
        fr_ = base[(np.arange(chunk) + k) % 4]
        o = cas.step(fr_, replay=[boxes(k * chunk + i) for i in range(chunk)])
        ids |= {r[0] for f in o["tracks"] for r in f}
        ctx.synchronize()
        used.append(device_bytes_in_use())
        live.append(len(cas.persons.streams))
    cas.flush()
    assert len(ids) >= 3, ids                             # the scenario did create and retire tracks
    assert max(live) <= 3, live                           # a retired track's stream is closed once its last 3D frames left
    assert not cas.persons.streams
    assert abs(used[-1] - used[3]) <= 2 << 20, [u >> 20 for u in used]
    assert max(used[3:]) - min(used[3:]) <= 8 << 20, [u >> 20 for u in used]


def test_detector_enqueue_collect_and_lookahead_modes(ctx):
    """pp_detector_enqueue / pp_detector_collect (ABI 10) == pp_detector_run bit for bit, one pass in flight at a time (a second enqueue
    is refused), proposals on request; and a cascade gives the same tracks / key points / 3D in every look-ahead mode -- "stream" (the
    next chunk's pass enqueued on the detector's own context while this chunk's stages run), "pipeline" (same stream), "off"."""
    from posepipeline_amd import _lib as L
    from posepipeline_amd.cascade import Cascade, collect
    det_sd, pose_spec, pose_sd, lift_sd = _setup(135, 240)
    rng = np.random.default_rng(17)
    frames = np.stack([synth_frame(rng, 135, 240) for _ in range(8)])
    det = fr.Detector(ctx, det_sd, 135, 240, max_frames=4, numerics="exact")
    ref, ref_p = det.run(frames[:4], want_proposals=True)
    det.enqueue(frames[:4], want_proposals=True)
    with pytest.raises(L.PosePipeHipError, match="has not been collected"):
        det.enqueue(frames[4:])
    got, got_p = det.collect()
    assert all(np.array_equal(a, b) for a, b in zip(got, ref)) and all(np.array_equal(a, b) for a, b in zip(got_p, ref_p))
    with pytest.raises(L.PosePipeHipError, match="no pass in flight"):
        det.collect()
    det.enqueue(frames[4:])
    assert all(np.array_equal(a, b) for a, b in zip(det.collect(), det.run(frames[4:])))
    outs = {}
    for mode in ("stream", "pipeline", "off"):
        cas = Cascade(ctx, det_sd, pose_sd, lift_sd, 135, 240, chunk=4, max_persons=2, pose_spec=pose_spec, overlap_detector=mode, numerics="exact")
        assert cas.lookahead == mode and (cas.det_ctx is not cas.ctx) == (mode == "stream")
        res = [cas.step(frames[:4], prefetch=(frames[4:], None)), cas.step(frames[4:]), cas.flush()]
        outs[mode] = ([t for o in res for t in o["tracks"]], collect(res, "keypoints"), collect(res, "keypoints_3d"))
        # a prefetch that is never consumed (the caller stops, or asks for another chunk) is drained, not leaked
        cas.reset()
        cas.step(frames[:4], prefetch=(frames[4:], None))
        o2 = cas.step(frames[:4])            # NOT the prefetched chunk: its pass is collected and dropped, this one runs now
        assert [r[0] for fr_ in o2["tracks"] for r in fr_] is not None
        cas.release()
    for mode in ("pipeline", "off"):
        assert outs[mode][0] == outs["stream"][0]
        for a, b in ((outs[mode][1], outs["stream"][1]), (outs[mode][2], outs["stream"][2])):
            assert sorted(a) == sorted(b)
            for tid in a:
                assert a[tid][0] == b[tid][0] and np.array_equal(a[tid][1], b[tid][1])
