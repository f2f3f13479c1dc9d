"""The whole hot path in one object: detect -> track -> select person boxes -> top-down 2D -> 3D lifting.

This is what the three Computed tables of the reference do one after the other for a video
(`TrackingBbox.make` -> `PersonBbox.make` -> `TopDownPerson.make` -> `LiftingPerson.make`,
pose_pipeline/pipeline.py:515-578, 656-687, 1017-1095, 1259-1416, driven by
utils/standard_pipelines.py:110-164), restructured as a chunked stream: a chunk of frames is resident
on the device once and every stage consumes it there, instead of three full decodes of the file and
batch-1 model calls.  Used by bench.py (headline metric) and by tests; the table-by-table drop-in path
lives in posepipeline_amd/wrappers/.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from . import ops
from .models import faster_rcnn as fr
from .models import hrnet
from .models import vitpose
from .models import videopose3d as vp3d
from .program import Net
from .tracking import Tracker
from .wrappers.videopose3d import lift, normalize_screen_coordinates


class Cascade:
    """tracking: "MMTrack_deepsort" (Faster-RCNN R50-FPN + SORT, det_sd = detector weights) or "DeepSortYOLOv4"
    (tracking_method 0, the reference recipes' default: det_sd = (yolov4 weights, mars-small128 weights))."""

    def __init__(self, ctx: L.Context, det_sd, pose_sd: dict, lift_sd: dict, src_h: int, src_w: int,
                 chunk: int = 8, max_persons: int = 1, pose_spec=None, post="unbiased", blur_kernel=17,
                 tracking: str = "MMTrack_deepsort"):
        self.ctx = ctx
        self.src = (src_h, src_w)
        self.chunk = chunk
        self.max_persons = max_persons
        self.tracking = tracking
        if tracking == "DeepSortYOLOv4":
            from .models import mars, yolov4
            self.detector = yolov4.YoloV4Detector(ctx, det_sd[0], src_h, src_w, max_frames=chunk)
            self.encoder = mars.MarsEncoder(ctx, det_sd[1], src_h, src_w, max_patches=max(64, chunk * max_persons))
        else:
            assert tracking == "MMTrack_deepsort", tracking
            self.detector = fr.Detector(ctx, det_sd, src_h, src_w, max_frames=chunk)
        self.pose_spec = pose_spec or hrnet.hrnet_w48_384x288()
        if isinstance(self.pose_spec, vitpose.VitPoseSpec):     # BASELINE.json configs[4]: ViTPose 2D stage (UDP, bf16 MFMA)
            pose_prog = vitpose.build_vitpose_program(self.pose_spec, pose_sd)
            post, blur_kernel, shift = "udp", 11, False
        else:
            pose_prog = hrnet.build_hrnet_program(self.pose_spec, pose_sd)
            shift = True
        self.pose_net = Net(ctx, pose_prog, max_batch=2 * chunk * max_persons)
        self.topdown = ops.TopDown(self.pose_net, 17, flip_perm=hrnet.flip_perm(17), shift_heatmap=shift, post=post,
                                   blur_kernel=blur_kernel)
        self.lift_spec = vp3d.VideoPose3DSpec()
        self.lift_net = Net(ctx, vp3d.build_videopose3d_program(self.lift_spec, lift_sd), max_batch=max(1, max_persons))
        self.reset()

    def reset(self):
        if self.tracking == "DeepSortYOLOv4":
            self.tracker = Tracker(mode=0, feat_dim=128, max_cosine_distance=0.3)       # parser.py:35-47
        else:
            self.tracker = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
        self.tracks = []          # per frame: list of (track_id, x1, y1, x2, y2, score)
        self.kp2d = {}            # track_id -> list of (frame, (17,3))

    @property
    def flops_per_frame(self):
        """algorithmic conv FLOPs per frame at max_persons persons (detector + 2 x HRNet per person + lifting)"""
        return (self.detector.flops_per_frame + 2 * self.max_persons * self.pose_net.prog.flops +
                self.max_persons * self.lift_net.prog.flops / self.lift_spec.chunk)

    def step(self, frames, frames_dev=None, replay=None):
        """One chunk.  frames: numpy [B][H][W][3] u8 BGR, or frames_dev=(device pointer, B).
        replay: optional per-frame [n][5] boxes that stand in for the detector's output downstream (bench
        with random-weight detectors) -- the detector still runs.
        Returns dict(tracks=per-frame rows, keypoints={track_id: (B,17,3)}, keypoints_3d={track_id: (B,17,3)})."""
        b = frames_dev[1] if frames_dev is not None else frames.shape[0]
        dets = self.detector.run(frames, frames_dev=frames_dev)
        chunk_tracks = []
        if self.tracking == "DeepSortYOLOv4":
            # wrappers/deep_sort_yolov4/parser.py:52-86 per frame: persons -> appearance features -> NMS(1.0) -> DeepSORT;
            # every live track is reported (tentative and missed ones with their Kalman box), like the reference's tables
            if replay is not None:      # [n][5] x1 y1 x2 y2 score -> the int (x, y, w, h) boxes yolo.detect_image returns
                dets = [(np.array([[int(r[0]), int(r[1]), int(r[2] - r[0]), int(r[3] - r[1])] for r in rows], np.int64).reshape(-1, 4),
                         np.asarray(rows, np.float32).reshape(-1, 5)[:, 4]) for rows in replay]
            feats = self.encoder.encode(frames, [bx for bx, _ in dets], frames_dev=frames_dev)
            for (boxes, conf), feat in zip(dets, feats):
                tlwh, sc = boxes.astype(np.float64), conf.astype(np.float64)
                keep = ops.nms(self.ctx, tlwh, sc, 1.0, convention=1) if len(tlwh) else np.zeros(0, np.int64)
                ids, t, _ = self.tracker.step(tlwh[keep], sc[keep], feat[keep])
                chunk_tracks.append([(int(i), bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3], 1.0) for i, bb in zip(ids, t)])
        else:
            if replay is not None:
                dets = replay
            for rows in dets:
                rows = np.asarray(rows, np.float32).reshape(-1, 5)
                ids, _, info = self.tracker.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
                chunk_tracks.append([(int(i), *rows[j]) for i, j in zip(ids, info[:, 1])])
        f0 = len(self.tracks)
        self.tracks += chunk_tracks
        # person-frames for the 2D stage: every tracked box of the chunk (up to max_persons per frame)
        fidx, boxes, owner = [], [], []
        for t, fr_tracks in enumerate(chunk_tracks):
            for (tid, x1, y1, x2, y2, _s) in fr_tracks[: self.max_persons]:
                fidx.append(t)
                boxes.append([x1, y1, x2 - x1, y2 - y1])
                owner.append(tid)
        kp = {}
        kp3d = {}
        if boxes:
            if frames_dev is not None:
                k2, _ = self.topdown.run(frames_dev[0], np.array(fidx, np.int32), np.array(boxes, np.float64),
                                         frames_dev_shape=(b, self.src[0], self.src[1]))
            else:
                k2, _ = self.topdown.run(frames, np.array(fidx, np.int32), np.array(boxes, np.float64))
            for i, tid in enumerate(owner):
                self.kp2d.setdefault(tid, []).append((f0 + fidx[i], k2[i]))
            for tid in sorted(set(owner)):
                hist = self.kp2d[tid]
                # lift this chunk's frames with the context accumulated so far (halo = receptive field)
                n_new = sum(1 for fi, _ in hist if fi >= f0)
                ctx_frames = hist[-(n_new + 2 * self.lift_spec.pad):]
                arr = np.stack([k for _, k in ctx_frames])
                kn = normalize_screen_coordinates(arr[:, :, :2].astype(np.float64), self.src[1], self.src[0])
                out = lift(self.lift_net, self.lift_spec, kn)
                kp[tid] = arr[-n_new:]
                kp3d[tid] = out[-n_new:]
        return dict(tracks=chunk_tracks, keypoints=kp, keypoints_3d=kp3d)

    def run_video(self, video, replay_fn=None, max_frames=None):
        """Whole clip, read once: frames stream through page-locked staging buffers and the copy stream
        (streaming.FrameStreamer) while the previous chunk computes.  video: video.open_video() object.
        replay_fn(first, n) -> per-frame replay boxes (see step).  Yields step() results, one per chunk."""
        from .streaming import FrameStreamer
        assert (video.height, video.width) == self.src, ((video.height, video.width), self.src)
        streamer = FrameStreamer(self.ctx, video, self.chunk, max_frames=max_frames)
        try:
            for dev_ptr, n, first in streamer:
                out = self.step(None, frames_dev=(dev_ptr, n), replay=None if replay_fn is None else replay_fn(first, n))
                streamer.release()
                out["first_frame"] = first
                yield out
        finally:
            streamer.close()
