"""Drop-in for pose_pipeline/wrappers/mmtrack.py:8-62 `mmtrack_bounding_boxes`.

Same signature and the same return structure: one list per decoded frame of
    {"track_id": int, "tlbr": ndarray(4,) x1y1x2y2, "tlhw": ndarray(4,) [x1, y1, w, h], "confidence": float}
(the key really is spelled `tlhw`, :50-60).  The reference calls `mmtrack.apis.inference_mot` once per
frame; here the clip is streamed to the device in batches, the detector runs batched on the GPU and the strictly
sequential association runs on the host.

Built:
  "deepsort"   mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py, the configuration the reference selects
               (wrappers/mmtrack.py:16-19): Faster-RCNN R50-FPN (pp_detector) + mmtrack SortTracker WITH its ReID
               appearance branch -- ResNet-50 ReID model on 256x128 crops of the detector's input tensor
               (models/reid_r50.py), Kalman gating, appearance then IoU assignment (tracking.SortReidTracker; the
               NaN-gated assignment is given an explicit reading there).  POSEPIPE_MMTRACK_REID=0 drops the appearance
               branch (= the sort_faster-rcnn configuration of the same directory, pp_tracker mode 1);
  "bytetrack"  YOLOX-X (models/yolox.py) + mmtrack ByteTracker (tracking.ByteTracker) of
               mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private.py -- restated from mmdet / mmtrack 0.x, unpinned.
tracktor / qdtrack use other model families and raise NotImplementedError.  Unknown names raise Exception like the
reference (:28-29).
"""
from __future__ import annotations

import numpy as np

import os

from .. import _lib, weights
from ..models import faster_rcnn as fr
from ..tracking import ByteTracker, SortReidTracker, Tracker
from ..video import open_video

BATCH = 16
BATCH_YOLOX = 4          # 800 x 1440 inputs: 1.4 GB of activations per frame
_KNOWN = ("tracktor", "deepsort", "bytetrack", "qdtrack")
_cache: dict = {}


def _detector(src_h, src_w, device=0, method="deepsort"):
    key = (src_h, src_w, device, method)
    if key not in _cache:
        ctx = _lib.Context(device)
        if method == "bytetrack":
            from ..models import yolox
            # init_cfg of mot/bytetrack/*-private-half.py:16-20: the COCO YOLOX-X checkpoint of mmdetection
            rel = "mmtracking/checkpoints/yolox_x_8x8_300e_coco_20211126_140254-1ef88d67.pth"
            sd = weights.get_state_dict(rel, yolox.yolox_param_shapes(), seed=6)
            if not os.path.exists(os.path.join(weights.model_data_dir(), rel)):
                yolox.seed_synthetic_head(sd)           # seeded weights: keep the candidate count realistic
            _cache[key] = (ctx, yolox.YoloXDetector(ctx, sd, src_h, src_w, max_frames=BATCH_YOLOX))
        else:
            sd = weights.get_state_dict("mmtracking/checkpoints/faster-rcnn_r50_fpn_4e_mot17-half-64ee2ed4.pth",
                                        fr.faster_rcnn_param_shapes(), seed=2)
            det = fr.Detector(ctx, sd, src_h, src_w, max_frames=BATCH)
            if method == "deepsort" and os.environ.get("POSEPIPE_MMTRACK_REID", "1") != "0":
                from ..models import reid_r50
                # init_cfg of the config's reid section (:39-42)
                rsd = weights.get_state_dict("mmtracking/checkpoints/tracktor_reid_r50_iter25245-a452f51f.pth",
                                             reid_r50.reid_param_shapes(), seed=7)
                det.reid = reid_r50.ReidEncoder(ctx, rsd, det)
            _cache[key] = (ctx, det)
    return _cache[key]


def _rows_to_dicts(track_results):
    return [
        {
            "track_id": int(x[0]),
            "tlbr": x[1:5],
            "tlhw": np.array([x[1], x[2], x[3] - x[1], x[4] - x[2]]),
            "confidence": x[5],
        }
        for x in track_results
    ]


def mmtrack_bounding_boxes(file_path, method="tracktor"):
    if method not in _KNOWN:
        raise Exception(f"Unknown config file for MMTrack method {method}")
    if method not in ("deepsort", "bytetrack"):
        raise NotImplementedError(f"MMTrack method {method!r}: only the Faster-RCNN + SORT and YOLOX + ByteTrack families "
                                  "are built (see module docstring)")

    from ..streaming import FrameStreamer
    cap = open_video(file_path)
    video_length = int(cap.num_frames)
    ctx, det = _detector(cap.height, cap.width, method=method)
    byte = method == "bytetrack"
    reid = getattr(det, "reid", None)
    tracker = ByteTracker() if byte else SortReidTracker() if reid is not None else Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)
    batch = BATCH_YOLOX if byte else BATCH

    tracks = []
    if video_length <= 0:
        cap.release()
        return tracks
    # the clip is read once and streamed to the device `batch` frames at a time (the reference: one cap.read() and one
    # blocking upload per frame, :38-45); a read failure simply ends the stream (:41-42)
    streamer = FrameStreamer(ctx, cap, min(batch, video_length), max_frames=video_length)
    try:      # an error in a stage must not leak the reader thread, the page-locked staging buffers and the open video
        for dev_ptr, n, _first in streamer:
            per_frame = det.run(None, frames_dev=(dev_ptr, n))          # [n][5] float32: x1 y1 x2 y2 score
            if reid is not None:
                # appearance embeddings of the detections the tracker keeps, from the detector's resident input tensor
                per_frame = [rows[tracker.keep(rows)] for rows in per_frame]
                embeds = reid.encode(per_frame)
            streamer.release()
            for k, rows in enumerate(per_frame):
                if reid is not None:
                    track_results = list(tracker.step(rows, embeds[k]))                          # [id, x1, y1, x2, y2, score]
                elif byte:
                    track_results = list(tracker.step(rows))
                else:
                    ids, _, info = tracker.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
                    track_results = [np.concatenate([[np.float32(i)], rows[j]]).astype(np.float32) for i, j in zip(ids, info[:, 1])]
                tracks.append(_rows_to_dicts(track_results))
    finally:
        streamer.close()
        cap.release()
    return tracks
