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
    monkeypatch.setenv("POSEPIPE_SYNTHETIC_WEIGHTS", "1")
    from posepipeline_amd import video
    from posepipeline_amd.wrappers import mmtrack as wmt
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


def test_cascade_chunks_match_oracle_chain(ctx):
    from posepipeline_amd.cascade import Cascade
    from posepipeline_amd.wrappers.videopose3d import normalize_screen_coordinates
    rng = np.random.default_rng(3)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(4)])
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_spec = hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=1)
    lift_spec = vp3d.VideoPose3DSpec()
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(lift_spec), seed=3)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    # replayed person boxes (random-weight detector): one person drifting right
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(4)]
    out = [cas.step(frames[0:2], replay=gt[0:2]), cas.step(frames[2:4], replay=gt[2:4])]
    assert [len(t) for o in out for t in o["tracks"]] == [1, 1, 1, 1]
    tid = out[0]["tracks"][0][0][0]
    assert all(t[0][0] == tid for o in out for t in o["tracks"])                      # one identity throughout
    k2 = np.concatenate([o["keypoints"][tid] for o in out])
    boxes = np.array([[g[0, 0], g[0, 1], g[0, 2] - g[0, 0], g[0, 3] - g[0, 1]] for g in gt], np.float64)
    ref2 = oracle_topdown(pose_sd, 32, frames, boxes, (96, 128), "unbiased", 17)
    for i in range(4):
        assert np.abs(k2[i][:, :2] - ref2[i][:, :2]).max() <= 1e-3
        assert np.array_equal(k2[i][:, 2], ref2[i][:, 2].astype(np.float32))
    # lifting of the second chunk sees frames 0..3 as context (streaming, edge-replicated)
    kn = normalize_screen_coordinates(k2[:, :, :2].astype(np.float64), w, h).astype(np.float32)
    ref3 = onets.VideoPose3DRef(lift_sd).forward(onets.videopose3d_windows(kn, 121))
    assert np.array_equal(out[1]["keypoints_3d"][tid], ref3[2:4])
    # and the detector really ran on the chunk: its own boxes equal the oracle's
    d0 = cas.detector.run(frames[:1])[0]
    assert np.array_equal(d0, odet.detect(odet.FasterRCNNRef(det_sd), frames[0][:, :, ::-1]))


def test_streamed_video_equals_host_chunks(ctx, tmp_path):
    """run_video (reader thread -> page-locked staging -> copy stream, ragged last chunk) returns exactly what
    step() returns on the same frames passed from host memory."""
    from posepipeline_amd import video
    from posepipeline_amd.cascade import Cascade
    rng = np.random.default_rng(5)
    h, w = 135, 240
    frames = np.stack([synth_frame(rng, h, w) for _ in range(5)])
    path = str(tmp_path / "clip.ppvid")
    video.write_ppvid(path, frames)
    det_sd = synth.synth_state_dict(fr.faster_rcnn_param_shapes(), seed=2)
    pose_spec = hrnet.HRNetSpec(32, 17, 128, 96)
    pose_sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(pose_spec), seed=1)
    lift_sd = synth.synth_state_dict(vp3d.videopose3d_param_shapes(vp3d.VideoPose3DSpec()), seed=3)
    cas = Cascade(ctx, det_sd, pose_sd, lift_sd, h, w, chunk=2, max_persons=1, pose_spec=pose_spec)
    gt = [np.array([[60 + 4 * t, 20, 130 + 4 * t, 120, 0.9]], np.float32) for t in range(5)]
    ref = [cas.step(frames[i:i + 2], replay=gt[i:i + 2]) for i in (0, 2, 4)]
    cas.reset()
    got = list(cas.run_video(video.open_video(path), replay_fn=lambda first, n: gt[first:first + n]))
    assert [o["first_frame"] for o in got] == [0, 2, 4]
    for a, b in zip(ref, got):
        assert a["tracks"] == b["tracks"]
        assert a["keypoints"].keys() == b["keypoints"].keys()
        for tid in a["keypoints"]:
            assert np.array_equal(a["keypoints"][tid], b["keypoints"][tid])
            assert np.array_equal(a["keypoints_3d"][tid], b["keypoints_3d"][tid])
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
    out = [cas.step(frames[0:2], replay=gt[0:2]), cas.step(frames[2:4], replay=gt[2:4])]
    tracks = [t for o in out for t in o["tracks"]]
    assert [len(t) for t in tracks] == [1, 1, 1, 1] and len({t[0][0] for t in tracks}) == 1     # one identity (id 1)
    assert tracks[0][0][0] == 1                                                                 # deep_sort ids start at 1
    tid = tracks[0][0][0]
    k2 = np.concatenate([o["keypoints"][tid] for o in out])
    boxes = np.array([[t[0][1], t[0][2], t[0][3] - t[0][1], t[0][4] - t[0][2]] for t in tracks], np.float64)
    # first frame: the Kalman mean is initialised from the detection, so the track box is the (int-truncated) input box
    assert np.allclose(boxes[0], [60, 20, 70, 100])
    ref2 = oracle_topdown(pose_sd, 32, frames, boxes, (96, 128), "unbiased", 17)
    for i in range(4):
        assert np.abs(k2[i][:, :2] - ref2[i][:, :2]).max() <= 1e-3
    assert out[1]["keypoints_3d"][tid].shape == (2, 17, 3)
