"""The whole hot path in one object: detect -> track -> select person boxes -> top-down 2D -> 3D lifting.

This is what the Computed tables of the reference do one after the other for a video
(`TrackingBbox.make` -> `PersonBbox.make` -> `TopDownPerson.make` -> `LiftingPerson.make`,
pose_pipeline/pipeline.py:515-578, 656-687, 1017-1095, 1259-1416, driven by
utils/standard_pipelines.py:110-164), restructured as a chunked stream: a chunk of frames is resident
on the device once and every stage consumes it there, instead of three full decodes of the file and
batch-1 model calls.  Used by bench.py (headline metric) and by tests; the table-by-table drop-in path
lives in posepipeline_amd/wrappers/.

What is emitted is defined in person_stream.py: for every followed track id the values the reference's table chain
stores for `keep_tracks = [tid]` -- PersonBbox's bfill / ffill(limit 2), zero rows for absent frames, VideoPose3D
over the window [t-121, t+121] of the WHOLE clip -- with the latency those look-aheads need: the crops of frames whose
box is back-filled from the next chunk come from a 2-frame tail buffer kept on the device, frame t is lifted once frame
t+121 has its key points, `flush()` emits the rest at the end of the clip.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib as L
from . import ops
from .models import faster_rcnn as fr
from .models import hrnet
from .models import vitpose
from .models import videopose3d as vp3d
from .person_stream import FILL_LIMIT, PersonStreams, collect  # noqa: F401  (collect: re-exported for callers)
from .program import Net
from .tracking import Tracker
from .wrappers.videopose3d import lift


# Default certification thresholds of Cascade(id_numerics="certified"): 4x the largest deviation between the fp16-form and the
# float32-MFMA detector measured per quantity on 1080p frames (tools/margin_probe.py, profiles/r06_margin_probe.txt: RPN scores
# 5.2e-6, proposal coordinates 2.9e-6 of the box size -> IoU ~1e-5, RoI scores 1.2e-5, detection boxes 3.1e-6 of their size,
# detection scores 4.5e-6), in the units of each margin (include/posepipe_hip.h).
CERTIFY_EPS = {"rpn_cut": 2e-5, "rpn_nms": 5e-5, "rpn_top": 2e-5, "roi_level": 2e-5, "score_thr": 5e-5, "det_nms": 5e-5,
               "det_top": 5e-5, "det_order": 4e-5, "trk_score": 2e-5}


# HIP streams ("lanes") per program inside a cascade.  Filled in from the same-box A/B of round 6 (profiles/r06_lanes_ab.txt).
LANES = {"detector": 2, "pose": 2}


class Cascade:
    """tracking: "MMTrack_deepsort" (Faster-RCNN R50-FPN, det_sd = detector weights; association = mmtrack SortTracker: with
    reid_sd (ReID ResNet-50 weights) the DeepSORT configuration with its appearance branch, without it the SORT
    configuration BASELINE.json configs[2] names) or "DeepSortYOLOv4" (tracking_method 0, the reference recipes' default:
    det_sd = (yolov4 weights, mars-small128 weights)).
    max_persons / keep_tracks: which track ids are followed (person_stream.PersonStreams).
    In "DeepSortYOLOv4" mode every track the tracker keeps is a row of every frame (tentative and missed ones with their
    Kalman box), because that is what the reference stores (parser.py:76-86) and PersonBbox selects from."""

    tracker_score_thr = 0.5       # SortTracker obj_score_thr (mot/deepsort/*.py:43-54)

    def __init__(self, ctx: L.Context, det_sd, pose_sd: dict, lift_sd: dict, src_h: int, src_w: int,
                 chunk: int = 8, max_persons: int = 1, pose_spec=None, post="unbiased", blur_kernel=17,
                 tracking: str = "MMTrack_deepsort", keep_tracks=None, flip_pairs=None, blob_fn=None, reid_sd=None,
                 overlap_detector: bool | None = None, numerics=None, id_numerics=None, certify_eps=None, det_lanes=None, pose_lanes=None):
        """blob_fn(name, program) -> (device pointer, n_floats) or None, name in "det_a", "det_b", "pose", "lift" (called in
        that order): a weight blob that is already resident on the device -- parallel.broadcast_blob_device delivers rank
        0's over RCCL; the *_sd arguments then only define the program structure (ops, buffers, blob offsets).
        overlap_detector: the look-ahead of the detector pass -- None / True / "stream" (default): the detector gets its own context
        (HIP stream + lanes) and `step(..., prefetch=next chunk)` ENQUEUES its pass over the next chunk there (pp_detector_enqueue:
        asynchronous, no thread); the next step collects it.  "pipeline": the same on the cascade's one stream; False / "off": none.
        Results are unchanged (same kernels, same order per stage).  Not with the ReID branch (it reads the detector's resident input
        tensor of the CURRENT chunk) nor with certified ids (synchronous passes).  POSEPIPE_OVERLAP_DETECTOR=0 / 1 / 3 overrides
        (off / stream / pipeline)."""
        # numerics: "exact" / "split" / None (= the process default at this moment) for EVERY program this cascade creates, passed
        # down explicitly -- two threads building cascades in different modes do not interfere (unlike _lib.default_numerics).
        # id_numerics (round 5): numerics of the programs whose outputs feed INTEGER decisions -- the detector (top-k, NMS, score
        # thresholds -> which boxes exist, in which order) and the appearance encoder (gated assignment) -- where it should differ from
        # the rest.  numerics="split", id_numerics="exact" is the "integer-exact" configuration: track ids, bbox indices and `present`
        # are the oracle's BY CONSTRUCTION (the float32-MFMA detector is bit-identical to it, the tracker is host float64), while the
        # pose / lifting programs, whose outputs are held to 1e-3 px / mm, run on the fast kernels.
        # id_numerics="certified" (round 6): the detector runs on the fast kernels WITH decision margins (pp_detector_enable_margins);
        # a frame whose every margin clears `certify_eps` keeps its fast detections, the others are run again through a float32-MFMA
        # detector (bit-identical to the oracle).  Which frames were certified is reported per step (`certified`), the policy below
        # falls back to the exact detector alone while fewer than half of a chunk's frames certify.
        self.certified = id_numerics == "certified"
        if self.certified:
            assert tracking == "MMTrack_deepsort" and reid_sd is None, "certified ids: the Faster-RCNN + SORT configuration"
            id_numerics = "split"
        id_numerics = numerics if id_numerics is None else id_numerics
        self.ctx = ctx
        self.det_ctx = ctx
        self._pending = None          # chunk key of the detector pass enqueued by step(prefetch=...)
        self.det_timing = None        # per-stage times of the detector pass this step consumed
        self.src = (src_h, src_w)
        self.chunk = chunk
        self.max_persons = max_persons
        self.tracking = tracking
        self.keep_tracks = keep_tracks
        blob_fn = blob_fn or (lambda name, prog: None)
        self.reid = None
        if tracking == "DeepSortYOLOv4":
            from .models import mars, yolov4
            self.detector = yolov4.YoloV4Detector(ctx, det_sd[0], src_h, src_w, max_frames=chunk, numerics=id_numerics)
            self.encoder = mars.MarsEncoder(ctx, det_sd[1], src_h, src_w, max_patches=max(64, chunk * max_persons), numerics=id_numerics)
        else:
            assert tracking == "MMTrack_deepsort", tracking
            # look-ahead mode (see the docstring).  Same-box (profiles/r06_lookahead_modes.txt): off 580, pipeline 588, stream 610 frames/s
            # (the worker-thread form of rounds 3 - 5: 595 against 600 for stream).  What pays is the SECOND QUEUE: the kernel trace
            # (profiles/r06_overlap_lookahead_stream.txt) shows the next chunk's detector kernels executing beside this chunk's pose kernels
            # for 18 - 35 % of the pose stage's kernel time; on one stream nothing co-runs and only the host-side gaps (~1 %) are recovered.
            env = os.environ.get("POSEPIPE_OVERLAP_DETECTOR")
            if env is not None:
                overlap_detector = {"0": "off", "1": "stream", "3": "pipeline"}.get(env, "stream")
            elif overlap_detector is None or overlap_detector is True:
                overlap_detector = "stream"
            elif overlap_detector is False:
                overlap_detector = "off"
            if reid_sd is not None or self.certified:
                overlap_detector = "off"
            self.lookahead = overlap_detector
            self.det_ctx = L.Context(ctx.device) if overlap_detector == "stream" else ctx
            self.detector = fr.Detector(self.det_ctx, det_sd, src_h, src_w, max_frames=chunk, blob_fn=blob_fn, numerics=id_numerics)
            if self.certified:
                self.certify_eps = dict(CERTIFY_EPS if certify_eps is None else certify_eps)
                self.detector.enable_margins(True, self.certify_eps["rpn_nms"] / (2.0 * self.certify_eps["rpn_cut"]))
                # (built from det_sd itself: blob_fn's broadcast protocol hands out every program's blob once, in a fixed order)
                self.detector_exact = fr.Detector(self.det_ctx, det_sd, src_h, src_w, max_frames=chunk, numerics="exact")
                self._gather_dev = None           # device scratch: the uncertified frames of a chunk, contiguous
                self._exact_only = False          # policy state: skip the fast pass while certification keeps failing
                self._since_probe = 0
                self.certify_stats = {"frames": 0, "certified": 0, "exact_only_frames": 0}
            if reid_sd is not None:
                from .models import reid_r50
                self.reid = reid_r50.ReidEncoder(ctx, reid_sd, self.detector, max_crops=max(64, chunk * max_persons), blob_fn=blob_fn, numerics=id_numerics)
        self.pose_spec = pose_spec or hrnet.hrnet_w48_384x288()
        if isinstance(self.pose_spec, vitpose.VitPoseSpec):     # BASELINE.json configs[4]: ViTPose 2D stage (UDP, bf16 MFMA)
            pose_prog = vitpose.build_vitpose_program(self.pose_spec, pose_sd)
            post, blur_kernel, shift = "udp", 11, False
            self.k = 17
        else:
            pose_prog = hrnet.build_hrnet_program(self.pose_spec, pose_sd)
            shift = True
            self.k = int(self.pose_spec.num_joints)
        self.pose_net = Net(ctx, pose_prog, max_batch=2 * chunk * max_persons, blob_dev=blob_fn("pose", pose_prog), numerics=numerics)
        if flip_pairs is None:
            flip_pairs = {17: hrnet.COCO_FLIP_PAIRS, 133: hrnet.WHOLEBODY_FLIP_PAIRS, 136: hrnet.HALPE_FLIP_PAIRS}[self.k]
        self.topdown = ops.TopDown(self.pose_net, self.k, flip_perm=hrnet.flip_perm(self.k, flip_pairs), shift_heatmap=shift,
                                   post=post, blur_kernel=blur_kernel)
        self.lift_spec = vp3d.VideoPose3DSpec()
        lift_prog = vp3d.build_videopose3d_program(self.lift_spec, lift_sd)
        self.lift_net = Net(ctx, lift_prog, max_batch=max(1, max_persons), blob_dev=blob_fn("lift", lift_prog), numerics=numerics)
        # lanes per program (pp_net_set_lane_count; results identical for any count): measured round 6 on the 1080p cascade, see LANES
        env_det, env_pose = os.environ.get("POSEPIPE_DET_LANES"), os.environ.get("POSEPIPE_POSE_LANES")
        det_lanes = int(env_det) if env_det else (LANES["detector"] if det_lanes is None else det_lanes)
        pose_lanes = int(env_pose) if env_pose else (LANES["pose"] if pose_lanes is None else pose_lanes)
        if "POSEPIPE_NET_LANES" not in os.environ:
            for det in (getattr(self, "detector", None), getattr(self, "detector_exact", None)):
                if det is not None and hasattr(det, "net_a"):
                    det.net_a.set_lane_count(det_lanes)
            if not isinstance(self.pose_spec, vitpose.VitPoseSpec):
                self.pose_net.set_lane_count(pose_lanes)
        self.frame_bytes = src_h * src_w * 3
        self.tail_dev = None          # device copy of the last FILL_LIMIT frames (slot = frame % FILL_LIMIT)
        self.tail_host = None
        self._cur = None              # (frames, frames_dev, first frame) of the chunk being processed
        self.reset()

    def _drop_prefetch(self):
        if self._pending is not None:
            self._pending = None
            self.detector.collect()

    def close(self):
        self._drop_prefetch()
        if self.tail_dev is not None and getattr(self.ctx, "handle", None):
            self.ctx.free(self.tail_dev)
        self.tail_dev = None
        if getattr(self, "_gather_dev", None) is not None and getattr(self.det_ctx, "handle", None):
            self.det_ctx.free(self._gather_dev)
        self._gather_dev = None

    def release(self):
        """close() + give the device memory of every program back (arenas, weights): the object is unusable afterwards.  A cascade at
        8 persons per frame holds ~50 GB; a process that builds several (bench.py's mode legs) releases them as it goes."""
        self.close()
        for name in ("topdown", "detector_exact", "detector", "reid", "encoder"):
            obj = getattr(self, name, None)
            if obj is not None and hasattr(obj, "close"):
                obj.close()
        for det in (getattr(self, "detector", None), getattr(self, "detector_exact", None)):
            for net in (getattr(det, "net_a", None), getattr(det, "net_b", None)):
                if net is not None:
                    net.close()
        for name in ("pose_net", "lift_net"):
            net = getattr(self, name, None)
            if net is not None:
                net.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._drop_prefetch()
        if self.tracking == "DeepSortYOLOv4":
            self.tracker = Tracker(mode=0, feat_dim=128, max_cosine_distance=0.3)       # parser.py:35-47
        elif getattr(self, "reid", None) is not None:
            from .tracking import SortReidTracker
            self.tracker = SortReidTracker()
        else:
            self.tracker = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=self.tracker_score_thr)
        # the lifting network takes the 17 COCO joints; wider heads (Halpe-136 / WholeBody-133) start with them
        self.persons = PersonStreams(self.k, self.lift_spec.pad, self.src, self._topdown_jobs,
                                     lambda kn: lift(self.lift_net, self.lift_spec, kn[:, :17]),
                                     max_persons=self.max_persons, keep_tracks=self.keep_tracks)

    @property
    def n_frames(self):
        return self.persons.n_frames

    @property
    def flops_per_frame(self):
        """algorithmic conv FLOPs per frame at max_persons persons (detector + 2 x HRNet per person + lifting)"""
        return (self.detector.flops_per_frame + 2 * self.max_persons * self.pose_net.prog.flops +
                self.max_persons * self.lift_net.prog.flops / self.lift_spec.chunk)

    # ---- stage 1: detect + associate ------------------------------------------------------------------------
    @staticmethod
    def _chunk_key(frames, frames_dev):
        return ("dev", int(frames_dev[0]), int(frames_dev[1])) if frames_dev is not None else ("host", id(frames))

    def _detect(self, frames, frames_dev):
        """the detector's pass over this chunk: the prefetched one if step(prefetch=) of the previous call started it"""
        if self._pending is not None:
            key, self._pending = self._pending, None
            dets, timing = self.detector.collect(), self.detector.timing()
            if key == self._chunk_key(frames, frames_dev):
                self.det_timing = timing
                return dets
        dets, self.det_timing = self._det_job(frames, frames_dev)
        return dets

    def _det_job(self, frames, frames_dev):
        """detector pass + its per-stage HIP-event times"""
        if self.certified:
            return self._det_job_certified(frames, frames_dev)
        dets = self.detector.run(frames, frames_dev=frames_dev)
        timing = self.detector.timing() if hasattr(self.detector, "timing") else None
        return dets, timing

    def uncertified_frames(self, dets, margins):
        """indices of the frames of a fast detector pass whose integer decisions are NOT safe under the split kernels' error bounds:
        a device margin (Detector.MARGIN_NAMES) or the tracker's score threshold within certify_eps"""
        eps = np.array([self.certify_eps[k] for k in fr.Detector.MARGIN_NAMES], np.float32)
        bad = (margins <= eps[None, :]).any(axis=1)
        thr = self.tracker_score_thr
        for i, rows in enumerate(dets):
            if len(rows) and np.abs(np.asarray(rows)[:, 4] - thr).min() <= self.certify_eps["trk_score"]:
                bad[i] = True
        return np.flatnonzero(bad)

    def _det_job_certified(self, frames, frames_dev):
        b = frames_dev[1] if frames_dev is not None else frames.shape[0]
        st = self.certify_stats
        st["frames"] += b
        # policy: while fewer than half of a chunk's frames certify, the fast pass buys nothing (its cost + the exact pass over the
        # rest exceeds the exact pass alone): run the exact detector only, and probe the fast pass again every 8th chunk
        if self._exact_only and self._since_probe < 8:
            self._since_probe += 1
            st["exact_only_frames"] += b
            self.last_certified = np.zeros(b, bool)
            return self.detector_exact.run(frames, frames_dev=frames_dev), self.detector_exact.timing()
        self._since_probe = 0
        dets = self.detector.run(frames, frames_dev=frames_dev)
        timing = self.detector.timing()
        redo = self.uncertified_frames(dets, self.detector.margins(b))
        self.last_certified = np.ones(b, bool)
        self.last_certified[redo] = False
        st["certified"] += b - len(redo)
        self._exact_only = 2 * len(redo) > b
        if len(redo) == b:
            dets = self.detector_exact.run(frames, frames_dev=frames_dev)
        elif len(redo):
            if frames_dev is not None:
                if self._gather_dev is None:
                    self._gather_dev = self.det_ctx.malloc(self.chunk * self.frame_bytes)
                for k, i in enumerate(redo):
                    self.det_ctx.d2d(self._gather_dev + k * self.frame_bytes, frames_dev[0] + int(i) * self.frame_bytes, self.frame_bytes)
                again = self.detector_exact.run(None, frames_dev=(self._gather_dev, len(redo)))
            else:
                again = self.detector_exact.run(np.ascontiguousarray(frames[redo]))
            for k, i in enumerate(redo):
                dets[int(i)] = again[k]
        return dets, timing

    def _prefetch(self, frames, frames_dev):
        if getattr(self, "lookahead", "off") == "off" or not hasattr(self.detector, "enqueue"):
            return
        self._held = frames                  # (host frames must outlive the pass)
        self.detector.enqueue(frames, frames_dev=frames_dev)
        self._pending = self._chunk_key(frames, frames_dev)

    def _track_chunk(self, frames, frames_dev, replay, dets):
        chunk_tracks = []
        if self.tracking == "DeepSortYOLOv4":
            # wrappers/deep_sort_yolov4/parser.py:52-86 per frame: persons -> appearance features -> NMS(1.0) -> DeepSORT
            if replay is not None:      # [n][5] x1 y1 x2 y2 score -> the int (x, y, w, h) boxes yolo.detect_image returns
                dets = [(np.array([[int(r[0]), int(r[1]), int(r[2] - r[0]), int(r[3] - r[1])] for r in rows], np.int64).reshape(-1, 4),
                         np.asarray(rows, np.float32).reshape(-1, 5)[:, 4]) for rows in replay]
            feats = self.encoder.encode(frames, [bx for bx, _ in dets], frames_dev=frames_dev)
            for (boxes, conf), feat in zip(dets, feats):
                tlwh, sc = boxes.astype(np.float64), conf.astype(np.float64)
                keep = ops.nms(self.ctx, tlwh, sc, 1.0, convention=1) if len(tlwh) else np.zeros(0, np.int64)
                ids, t, _ = self.tracker.step(tlwh[keep], sc[keep], feat[keep])
                chunk_tracks.append([(int(i), bb[0], bb[1], bb[0] + bb[2], bb[1] + bb[3], 1.0, bb.copy()) for i, bb in zip(ids, t)])
        elif self.reid is not None:
            # mmtrack DeepSORT: embeddings of the kept detections from the detector's resident input tensor, then
            # appearance + IoU association; ids survive misses (up to num_frames_retain), so liveness is the tracker's
            if replay is not None:
                dets = replay
            dets = [np.asarray(r, np.float32).reshape(-1, 5) for r in dets]
            dets = [r[self.tracker.keep(r)] for r in dets]
            embeds = self.reid.encode(dets)
            self._live_sets = []
            for rows, emb in zip(dets, embeds):
                out = self.tracker.step(rows, emb)
                chunk_tracks.append([(int(r[0]), *r[1:]) for r in out])
                self._live_sets.append(self.tracker.live_ids())
        else:
            if replay is not None:
                dets = replay
            for rows in dets:
                rows = np.asarray(rows, np.float32).reshape(-1, 5)
                ids, _, info = self.tracker.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
                chunk_tracks.append([(int(i), *rows[j]) for i, j in zip(ids, info[:, 1])])
        return chunk_tracks

    # ---- stage 2: top-down 2D on the decided person-frames --------------------------------------------------------
    def _topdown_jobs(self, jobs):
        """jobs: [(track_id, frame, tlwh)].  Crops come from the current chunk or, for frames of an earlier chunk whose box
        was back-filled only now, from the tail buffer."""
        frames, frames_dev, n0 = self._cur
        cap = max(1, self.pose_net.max_batch // 2)
        res = [None] * len(jobs)
        cur = [i for i, j in enumerate(jobs) if j[1] >= n0]
        old = [i for i, j in enumerate(jobs) if j[1] < n0]
        for group, is_tail in ((old, True), (cur, False)):
            for i0 in range(0, len(group), cap):
                part = group[i0:i0 + cap]
                boxes = np.array([jobs[i][2] for i in part], np.float64)
                if is_tail:
                    fidx = np.array([jobs[i][1] % FILL_LIMIT for i in part], np.int32)
                    if self.tail_dev is not None:
                        k2, _ = self.topdown.run(self.tail_dev, fidx, boxes, frames_dev_shape=(FILL_LIMIT, *self.src))
                    else:
                        k2, _ = self.topdown.run(self.tail_host, fidx, boxes)
                else:
                    fidx = np.array([jobs[i][1] - n0 for i in part], np.int32)
                    if frames_dev is not None:
                        k2, _ = self.topdown.run(frames_dev[0], fidx, boxes, frames_dev_shape=(frames_dev[1], *self.src))
                    else:
                        k2, _ = self.topdown.run(frames, fidx, boxes)
                for i, row in zip(part, k2):
                    res[i] = row
        return res

    def _save_tail(self, frames, frames_dev, n0, n1):
        """keep the last FILL_LIMIT frames: a frame that is absent now may be back-filled by the next chunk's boxes"""
        for t in range(max(n0, n1 - FILL_LIMIT), n1):
            slot = t % FILL_LIMIT
            if frames_dev is not None:
                if self.tail_dev is None:
                    self.tail_dev = self.ctx.malloc(FILL_LIMIT * self.frame_bytes)
                self.ctx.d2d(self.tail_dev + slot * self.frame_bytes, frames_dev[0] + (t - n0) * self.frame_bytes, self.frame_bytes)
            else:
                if self.tail_host is None:
                    self.tail_host = np.zeros((FILL_LIMIT, *self.src, 3), np.uint8)
                self.tail_host[slot] = frames[t - n0]

    def step(self, frames, frames_dev=None, replay=None, prefetch=None):
        """One chunk.  frames: numpy [B][H][W][3] u8 BGR, or frames_dev=(device pointer, B).
        replay: optional per-frame [n][5] boxes that stand in for the detector's output downstream (bench
        with random-weight detectors) -- the detector still runs.
        prefetch: (frames, frames_dev) of the NEXT chunk (already resident / valid until that step): its detector pass starts
        now and runs under this chunk's remaining stages (see overlap_detector).
        Returns dict(tracks = per-frame tracker rows of the chunk,
                     keypoints / keypoints_frames = {track_id: (n,K,3) / (n,) frame numbers} decided in this step,
                     keypoints_3d / keypoints_3d_frames = {track_id: (m,17,3) / (m,)} emitted in this step)."""
        b = frames_dev[1] if frames_dev is not None else frames.shape[0]
        self._live_sets = None
        dets_now = self._detect(frames, frames_dev)
        if prefetch is not None:
            self._prefetch(*prefetch)
        chunk_tracks = self._track_chunk(frames, frames_dev, replay, dets_now)
        n0 = self.persons.n_frames
        self.persons.ingest(chunk_tracks, self._live_sets)
        self._cur = (frames, frames_dev, n0)
        out = self.persons.advance(final=False)
        self._save_tail(frames, frames_dev, n0, n0 + b)
        self._cur = None
        out["tracks"] = chunk_tracks
        return out

    def flush(self):
        """End of clip: decide the open boxes (nothing follows), lift the remaining frames with the reference's edge
        replication.  Returns the same dict as step() with tracks = []."""
        self._cur = (None, None, self.persons.n_frames)
        out = self.persons.advance(final=True)
        self._cur = None
        out["tracks"] = []
        return out

    def run_video(self, video, replay_fn=None, max_frames=None, streamer=None):
        """Whole clip, read once: frames stream through page-locked staging buffers and the copy stream
        (streaming.FrameStreamer) while the previous chunk computes.  video: video.open_video() object.
        replay_fn(first, n) -> per-frame replay boxes (see step).  Yields step() results, one per chunk, then flush()'s.
        streamer: a FrameStreamer built by the caller over the same video (bench: staging-buffer allocation outside the timed
        window); it is closed here."""
        from .streaming import FrameStreamer
        assert (video.height, video.width) == self.src, ((video.height, video.width), self.src)
        if streamer is None:
            streamer = FrameStreamer(self.ctx, video, self.chunk, max_frames=max_frames)
        try:
            for dev_ptr, n, first in streamer:
                nxt = streamer.ahead          # chunk k + 1, already resident: the detector may start on it
                out = self.step(None, frames_dev=(dev_ptr, n), replay=None if replay_fn is None else replay_fn(first, n),
                                prefetch=None if nxt is None else (None, (nxt[0], nxt[1])))
                streamer.release()
                out["first_frame"] = first
                yield out
            out = self.flush()
            out["first_frame"] = self.n_frames
            yield out
        finally:
            streamer.close()
