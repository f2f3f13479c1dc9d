"""Drop-in for pose_pipeline/wrappers/mmtrack.py:8-62 `mmtrack_bounding_boxes`.

Same signature and the same return structure: one list per decoded frame of
    {"track_id": int, "tlbr": ndarray(4,) x1y1x2y2, "tlhw": ndarray(4,) [x1, y1, w, h], "confidence": float}
(the key really is spelled `tlhw`, :50-60).  The reference calls `mmtrack.apis.inference_mot` once per
frame; here frames are read in batches, the Faster-RCNN detector runs batched on the GPU (pp_detector) and
the strictly sequential association runs on the host in C++ (pp_tracker, mmtrack SortTracker semantics).

Built so far: the Faster-RCNN + SORT family.  `method="deepsort"` runs the detector and SortTracker of
mot/deepsort/*_faster-rcnn_fpn_4e_mot17-private-half.py WITHOUT the ReID appearance stage (that ResNet-50
ReID network is not built yet), i.e. the `sort_faster-rcnn` config; tracktor / bytetrack / qdtrack use
other detectors / trackers and raise NotImplementedError.  Unknown names raise Exception like the reference.
"""
from __future__ import annotations

import numpy as np

from .. import _lib, weights
from ..models import faster_rcnn as fr
from ..tracking import Tracker
from ..video import open_video

BATCH = 16
_KNOWN = ("tracktor", "deepsort", "bytetrack", "qdtrack")
_cache: dict = {}


def _detector(src_h, src_w, device=0):
    key = (src_h, src_w, device)
    if key not in _cache:
        sd = weights.get_state_dict("mmtracking/checkpoints/faster-rcnn_r50_fpn_4e_mot17-half-64ee2ed4.pth",
                                    fr.faster_rcnn_param_shapes(), seed=2)
        ctx = _lib.Context(device)
        _cache[key] = (ctx, fr.Detector(ctx, sd, src_h, src_w, max_frames=BATCH))
    return _cache[key]


def mmtrack_bounding_boxes(file_path, method="tracktor"):
    if method not in _KNOWN:
        raise Exception(f"Unknown config file for MMTrack method {method}")
    if method != "deepsort":
        raise NotImplementedError(f"MMTrack method {method!r}: only the Faster-RCNN + SORT family is built (see module docstring)")

    from ..streaming import FrameStreamer
    cap = open_video(file_path)
    video_length = int(cap.num_frames)
    ctx, det = _detector(cap.height, cap.width)
    tracker = Tracker(mode=1, match_iou_thr=0.5, obj_score_thr=0.5)

    tracks = []
    if video_length <= 0:
        cap.release()
        return tracks
    # the clip is read once and streamed to the device BATCH frames at a time (the reference: one cap.read() and one
    # blocking upload per frame, :38-45); a read failure simply ends the stream (:41-42)
    streamer = FrameStreamer(ctx, cap, min(BATCH, video_length), max_frames=video_length)
    for dev_ptr, n, _first in streamer:
        per_frame = det.run(None, frames_dev=(dev_ptr, n))          # [n][5] float32: x1 y1 x2 y2 score
        streamer.release()
        for rows in per_frame:
            ids, _, info = tracker.step(rows[:, :4].astype(np.float64), rows[:, 4].astype(np.float64))
            track_results = [np.concatenate([[np.float32(i)], rows[j]]).astype(np.float32) for i, j in zip(ids, info[:, 1])]
            tracks.append(
                [
                    {
                        "track_id": int(x[0]),
                        "tlbr": x[1:5],
                        "tlhw": np.array([x[1], x[2], x[3] - x[1], x[4] - x[2]]),
                        "confidence": x[5],
                    }
                    for x in track_results
                ]
            )
    streamer.close()
    cap.release()
    return tracks
