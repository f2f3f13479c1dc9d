"""Drop-in for pose_pipeline/wrappers/deep_sort_yolov4/parser.py:21-136 `tracking_bounding_boxes`
(tracking_method 0 `DeepSortYOLOv4`, pipeline.py:519-523; the default of every recipe in utils/standard_pipelines.py).

Same signature and return structure: one list per decoded frame with one dict per LIVE track (tentative and missed
ones included, parser.py:76-86):
    {"track_id": int, "tlhw": ndarray(4,) [x, y, w, h], "tlbr": ndarray(4,), "time_since_update": int}
The reference runs, per frame: YOLOv4 (`yolo.detect_image`, persons only) -> mars-small128 features of the boxes ->
`preprocessing.non_max_suppression(nms_max_overlap=1.0)` -> DeepSORT (max_cosine_distance 0.3, no budget).  Here frames are
read in batches; letterbox, the two networks and the box decode run on the GPU, NMS through pp_nms, the strictly
sequential association in C++ (pp_tracker mode 0, fixture-pinned against the reference's own deep_sort package).
`outfile` (annotated video writer, parser.py:29-31,88-129) is not supported.
Checkpoints (checkpoints_tf.py): mars-small128.pb (TensorFlow GraphDef) is read directly with a protobuf wire decoder,
yolo4.h5 (Keras/HDF5) through `tools/convert_yolo4_h5.py` -> yolo4.npz (or h5py when importable); without the files,
POSEPIPE_SYNTHETIC_WEIGHTS=1 runs seeded weights of the same architectures.
"""
from __future__ import annotations

import os

import numpy as np

from ... import _lib, ops, weights
from ...models import mars, yolov4
from ...tracking import Tracker
from ...video import open_video

BATCH = 16
_cache: dict = {}


def _params(relpath, shapes, seed, **kw):
    """the reference's checkpoint when it is installed (checkpoints_tf: GraphDef constants read directly, Keras HDF5 through
    its converted .npz or h5py), seeded weights of the same architecture with POSEPIPE_SYNTHETIC_WEIGHTS=1.
    Returns (params, seeded): seeded is True only for the synthetic branch -- the caller may adjust SEEDED weights
    (seed_person_head), never a real checkpoint, whichever file (.h5 / .pb or the converted .npz) it came from."""
    from ... import checkpoints_tf
    path = os.path.join(weights.model_data_dir(), relpath)
    converted = os.path.splitext(path)[0] + ".npz"
    if os.path.exists(path) or os.path.exists(converted):
        if relpath.endswith(".pb"):
            return checkpoints_tf.mars_params(path if os.path.exists(path) else converted, shapes), False
        return checkpoints_tf.yolo_params(path, shapes), False
    if os.environ.get("POSEPIPE_SYNTHETIC_WEIGHTS") == "1":
        return yolov4.synth_params(shapes, seed, **kw), True
    raise FileNotFoundError(f"{path} (set POSEPIPE_SYNTHETIC_WEIGHTS=1 to run with seeded synthetic weights)")


def _models(src_h, src_w, device=0):
    key = (src_h, src_w, device)
    if key not in _cache:
        ysd, seeded = _params("deep_sort_yolov4/yolo4.h5", yolov4.yolov4_param_shapes(), seed=4)
        if seeded:
            yolov4.seed_person_head(ysd)                 # seeded weights only: make the random head produce person candidates
        msd, _ = _params("deep_sort_yolov4/mars-small128.pb", mars.mars_param_shapes(), seed=5)
        ctx = _lib.Context(device)
        _cache[key] = (ctx, yolov4.YoloV4Detector(ctx, ysd, src_h, src_w, max_frames=BATCH),
                       mars.MarsEncoder(ctx, msd, src_h, src_w))
    return _cache[key]


def tracking_bounding_boxes(file_path, outfile=None):
    if outfile is not None:
        raise NotImplementedError("annotated video output (parser.py:29-31) is not part of the hot path")
    cap = open_video(file_path)
    video_length = int(cap.num_frames)
    ctx, yolo, encoder = _models(cap.height, cap.width)

    # Definition of the parameters (parser.py:35-47)
    max_cosine_distance = 0.3
    nms_max_overlap = 1.0
    tracker = Tracker(mode=0, feat_dim=128, max_cosine_distance=max_cosine_distance)

    from ...streaming import FrameStreamer
    tracks = []
    if video_length <= 0:
        cap.release()
        return tracks
    # the clip is read once and streamed to the device BATCH frames at a time; a read failure ends the stream (:52-53)
    streamer = FrameStreamer(ctx, cap, min(BATCH, video_length), max_frames=video_length)
    try:      # an error in a stage must not leak the reader thread, the page-locked staging buffers and the open video
        for dev_ptr, n, _first in streamer:
            dets = yolo.run(None, frames_dev=(dev_ptr, n))               # per frame: boxes [m][4] int, confidences [m]
            feats = encoder.encode(None, [b for b, _ in dets], frames_dev=(dev_ptr, n))
            streamer.release()
            for (boxes, conf), feat in zip(dets, feats):
                tlwh = boxes.astype(np.float64)                          # Detection.tlwh (detection.py:29)
                scores = conf.astype(np.float64)
                keep = ops.nms(ctx, tlwh, scores, nms_max_overlap, convention=1) if len(tlwh) else np.zeros(0, np.int64)
                ids, t_tlwh, info = tracker.step(tlwh[keep], scores[keep], feat[keep])
                tracks.append(
                    [
                        {
                            "track_id": int(i),
                            "tlhw": b.copy(),
                            "tlbr": np.concatenate([b[:2], b[:2] + b[2:]]),
                            "time_since_update": int(s[3]),
                        }
                        for i, b, s in zip(ids, t_tlwh, info)
                    ]
                )
    finally:
        streamer.close()
        cap.release()
    return tracks
