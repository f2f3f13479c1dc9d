"""Thin Python wrappers over the stage entry points of the C ABI (host-numpy or device-pointer arguments)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

# torchvision to_tensor + normalize constants of the top-down test pipeline
# (3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:132-136)
TOPDOWN_MEAN = (0.485, 0.456, 0.406)
TOPDOWN_STD = (0.229, 0.224, 0.225)


def normalize_lut(mean=TOPDOWN_MEAN, std=TOPDOWN_STD) -> np.ndarray:
    """[3][256] fp32 table of ((v/255) - mean[c]) / std[c], each step rounded to float32."""
    v = (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float32)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    return np.ascontiguousarray(((v[None, :] - m).astype(np.float32) / s).astype(np.float32))


def crop_affine_normalize(ctx: L.Context, frames: np.ndarray, frame_idx, bboxes, out_wh=(288, 384), lut=None,
                          chan_map=(0, 1, 2), flip=True, want_crop_u8=False):
    """frames [F][H][W][3] u8; bboxes [P][4] float64 TLWH (NaN row = absent).
    Returns dict(out=[P or 2P][out_h][out_w][4] fp32, center_scale=[P][4], valid=[P], crop_u8=...)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    f, h, w, c = frames.shape
    assert c == 3
    bboxes = np.ascontiguousarray(bboxes, dtype=np.float64).reshape(-1, 4)
    p = bboxes.shape[0]
    frame_idx = np.ascontiguousarray(frame_idx, dtype=np.int32)
    assert frame_idx.shape == (p,)
    lut = normalize_lut() if lut is None else np.ascontiguousarray(lut, np.float32)
    cm = np.asarray(chan_map, dtype=np.int32)
    ow, oh = out_wh
    out = np.empty((p * (2 if flip else 1), oh, ow, 4), dtype=np.float32)
    cs = np.zeros((p, 4), dtype=np.float32)
    valid = np.zeros((p,), dtype=np.int32)
    crop = np.empty((p, oh, ow, 3), dtype=np.uint8) if want_crop_u8 else None
    L.check(ctx.lib.pp_crop_affine_normalize(ctx.handle, L.ptr(frames), f, h, w, L.ptr(frame_idx), L.ptr(bboxes), p, ow, oh,
                                             L.ptr(lut), L.ptr(cm), int(flip), L.ptr(out), L.ptr(cs), L.ptr(crop),
                                             L.ptr(valid), L.PP_MEM_HOST), "pp_crop_affine_normalize")
    return dict(out=out, center_scale=cs, valid=valid, crop_u8=crop)


def flip_merge_decode(ctx: L.Context, hm: np.ndarray, hm_flip, center_scale, flip_perm=None, shift_heatmap=True,
                      post="unbiased", blur_kernel=17, want_merged=False):
    """hm, hm_flip [N][K][H][W] fp32 (host).  Returns (kpts [N][K][3], merged or None)."""
    hm = np.ascontiguousarray(hm, np.float32)
    n, k, h, w = hm.shape
    hf = None if hm_flip is None else np.ascontiguousarray(hm_flip, np.float32)
    perm = None if flip_perm is None else np.ascontiguousarray(flip_perm, np.int32)
    cs = np.ascontiguousarray(center_scale, np.float32).reshape(n, 4)
    kp = np.empty((n, k, 3), dtype=np.float32)
    merged = np.empty_like(hm) if want_merged else None
    post_i = {"unbiased": 1, "default": 0, None: -1}[post]
    L.check(ctx.lib.pp_flip_merge_decode(ctx.handle, L.ptr(hm), L.ptr(hf), n, k, h, w, L.ptr(perm), int(shift_heatmap),
                                         post_i, int(blur_kernel), L.ptr(cs), L.ptr(kp), L.ptr(merged), L.PP_MEM_HOST),
            "pp_flip_merge_decode")
    return kp, merged
