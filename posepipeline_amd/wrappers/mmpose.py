"""Drop-in for pose_pipeline/wrappers/mmpose.py:26-81 `mmpose_top_down_person`.

Same signature, same table reads (`PersonBbox.bbox`, the video of `key`), same return value
(ndarray (N, K, 3) = [x_px, y_px, score]; a frame whose bbox contains NaN yields zeros((K, 3)), which
makes the stacked result float64 exactly as in the reference, :67-69,:81).  What changes is the body:
the reference runs `inference_top_down_pose_model` once per frame at batch 1; here frames are read in
batches and each batch goes through pp_topdown (crop/normalise + mirrored copy -> HRNet on fp32 MFMA ->
flip-merge + DARK decode) on the GPU.

Channel order: the reference converts the BGR frame to RGB (:73) and mmpose's loader swaps it again,
so the network sees B in channel 0; with BGR frames from the reader that is the
identity channel map.
"""
from __future__ import annotations

import numpy as np

from .. import _lib, ops, weights
from ..models import hrnet, vitpose
from ..program import Net
from ..video import open_video

mmpose_joint_dictionary = {
    'MMPoseWholebody': ["Nose", "Left Eye", "Right Eye", "Left Ear", "Right Ear", "Left Shoulder", "Right Shoulder",
                        "Left Elbow", "Right Elbow", "Left Wrist", "Right Wrist", "Left Hip", "Right Hip", "Left Knee",
                        "Right Knee", "Left Ankle", "Right Ankle", "Left Big Toe", "Left Little Toe", "Left Heel",
                        "Right Big Toe", "Right Little Toe", "Right Heel"],
    'MMPoseHalpe': ["Nose", "Left Eye", "Right Eye", "Left Ear", "Right Ear", "Left Shoulder", "Right Shoulder",
                    "Left Elbow", "Right Elbow", "Left Wrist", "Right Wrist", "Left Hip", "Right Hip", "Left Knee",
                    "Right Knee", "Left Ankle", "Right Ankle", "Head", "Neck", "Pelvis", "Left Big Toe", "Right Big Toe",
                    "Left Little Toe", "Right Little Toe", "Left Heel", "Right Heel"],
    'MMPose': ["Nose", "Left Eye", "Right Eye", "Left Ear", "Right Ear", "Left Shoulder", "Right Shoulder",
               "Left Elbow", "Right Elbow", "Left Wrist", "Right Wrist", "Left Hip", "Right Hip", "Left Knee",
               "Right Knee", "Left Ankle", "Right Ankle"],
}

# method -> (spec factory, checkpoint under MODEL_DATA_DIR, K, flip pairs, post_process, blur kernel)
_METHODS = {
    "HRNet_W48_COCO": (hrnet.hrnet_w48_384x288, "mmpose/checkpoints/hrnet_w48_coco_384x288_dark-e881a4b6_20210203.pth",
                       17, hrnet.COCO_FLIP_PAIRS, "unbiased", 17),
    # same backbone, wider head + other flip pairs (wrappers/mmpose.py:41-52; configs
    # 3rdparty/mmpose/config/halpe/hrnet_w48_halpe_384x288_dark_plus.py, coco-wholebody/hrnet_w48_coco_wholebody_384x288_dark_plus.py)
    "HRNet_W48_HALPE": (hrnet.hrnet_w48_384x288, "mmpose/checkpoints/hrnet_w48_halpe_384x288_dark_plus-d13c2588_20211021.pth",
                        136, hrnet.HALPE_FLIP_PAIRS, "unbiased", 17),
    "HRNet_W48_COCOWholeBody": (hrnet.hrnet_w48_384x288,
                                "mmpose/checkpoints/hrnet_w48_coco_wholebody_384x288_dark-f5726563_20200918.pth",
                                133, hrnet.WHOLEBODY_FLIP_PAIRS, "unbiased", 17),
    # BASELINE.json configs[0-1] name the W32 256x192 member of the family (mmpose's plain W32 config decodes 'default')
    "HRNet_W32_COCO": (hrnet.hrnet_w32_256x192, "mmpose/checkpoints/hrnet_w32_coco_256x192-c78dce93_20200708.pth",
                       17, hrnet.COCO_FLIP_PAIRS, "default", 11),
    # BASELINE.json configs[4]; NOT a method of the reference wrapper (ViTPose is absent from /root/reference): the
    # published ViTPose COCO configs (UDP crop / DARK-UDP decode, modulate kernel 11, no heatmap shift), bf16 MFMA encoder
    "ViTPose_H_COCO": (vitpose.vitpose_huge, "mmpose/checkpoints/vitpose-h.pth", 17, hrnet.COCO_FLIP_PAIRS, "udp", 11),
    "ViTPose_L_COCO": (vitpose.vitpose_large, "mmpose/checkpoints/vitpose-l.pth", 17, hrnet.COCO_FLIP_PAIRS, "udp", 11),
    "ViTPose_B_COCO": (vitpose.vitpose_base, "mmpose/checkpoints/vitpose-b.pth", 17, hrnet.COCO_FLIP_PAIRS, "udp", 11),
}

BATCH = 32
_cache: dict = {}


def topdown_settings(method):
    """What `_model` hands ops.TopDown for `method` -- the test_cfg of the method's config (flip_test, post_process,
    shift_heatmap, modulate_kernel: hrnet_w48_coco_384x288_dark.py:81-85).  No device needed (tests/test_arch_configs.py)."""
    _, _, k, pairs, post, blur = _METHODS[method]
    return dict(num_joints=k, flip_perm=hrnet.flip_perm(k, pairs), shift_heatmap=post != "udp", post=post, blur_kernel=blur,
                chan_map=(0, 1, 2))


def _model(method, device=0):
    """(Context, Net, TopDown) for `method`, built once per process (the reference rebuilds per key, :57)."""
    if (method, device) not in _cache:
        if method not in _METHODS:
            # the reference has no else branch: pose_cfg is unbound for an unknown method
            raise UnboundLocalError(f"local variable 'pose_cfg' referenced before assignment (unknown method {method!r})")
        spec_fn, ckpt, k, pairs, post, blur = _METHODS[method]
        spec = spec_fn(k)
        ctx = _lib.Context(device)
        if isinstance(spec, vitpose.VitPoseSpec):
            sd = weights.get_state_dict(ckpt, vitpose.vitpose_param_shapes(spec), seed=1,
                                        synth=lambda shapes, seed: vitpose.synth_params(spec, seed))
            prog = vitpose.build_vitpose_program(spec, sd)
        else:
            sd = weights.get_state_dict(ckpt, hrnet.hrnet_param_shapes(spec), seed=1)
            prog = hrnet.build_hrnet_program(spec, sd)
        net = Net(ctx, prog, max_batch=2 * BATCH)
        td = ops.TopDown(net, **topdown_settings(method))
        _cache[(method, device)] = (ctx, net, td, k)
    return _cache[(method, device)]


def top_down_batches(td, num_keypoints, cap, bboxes, batch=BATCH):
    """Shared frame loop: the clip is read once and streamed to the device `batch` frames at a time (page-locked
    staging + copy stream, streaming.FrameStreamer) while the previous batch runs the fused stage; rows keep the
    reference's contract."""
    from ..streaming import FrameStreamer
    results = []
    n = len(bboxes)
    if n == 0:
        return results
    streamer = FrameStreamer(td.ctx, cap, min(batch, n), max_frames=n)
    try:
        for dev_ptr, m, first in streamer:
            bb = np.asarray(bboxes[first:first + m], dtype=np.float64)
            kp, valid = td.run(dev_ptr, np.arange(m, dtype=np.int32), bb, frames_dev_shape=(m, streamer.h, streamer.w))
            streamer.release()
            for j in range(m):
                if np.any(np.isnan(bb[j])):
                    results.append(np.zeros((num_keypoints, 3)))      # person not tracked in this frame (:67-69)
                else:
                    results.append(kp[j])
    finally:
        streamer.close()
    # should match the length of identified person tracks (wrappers/mmpose.py:63-64)
    assert len(results) == n, "video ended before the bbox track did"
    return results


def mmpose_top_down_person(key, method='HRNet_W48_COCO'):
    from ..pipeline import Video, PersonBbox

    _, _, td, num_keypoints = _model(method)
    bboxes = (PersonBbox & key).fetch1("bbox")
    video = Video.get_robust_reader(key, return_cap=False)
    cap = open_video(video)
    results = top_down_batches(td, num_keypoints, cap, bboxes)
    cap.release()
    return np.asarray(results)
