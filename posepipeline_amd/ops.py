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
                          chan_map=(0, 1, 2), flip=True, want_crop_u8=False, udp=False):
    """frames [F][H][W][3] u8; bboxes [P][4] float64 TLWH (NaN row = absent).
    udp: TopDownAffine(use_udp=True) transform (ViTPose configs) instead of the 3-point affine.
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
                                             L.ptr(lut), L.ptr(cm), int(bool(flip)) | (2 if udp else 0), L.ptr(out), L.ptr(cs), L.ptr(crop),
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
    post_i = {"unbiased": 1, "default": 0, "udp": 2, None: -1}[post]   # "udp": UDP crop + DARK-UDP decode (ViTPose)
    L.check(ctx.lib.pp_flip_merge_decode(ctx.handle, L.ptr(hm), L.ptr(hf), n, k, h, w, L.ptr(perm), int(shift_heatmap),
                                         post_i, int(blur_kernel), L.ptr(cs), L.ptr(kp), L.ptr(merged), L.PP_MEM_HOST),
            "pp_flip_merge_decode")
    return kp, merged


def nms(ctx: L.Context, boxes, scores, thr, convention=0):
    """convention 0: float32 x1y1x2y2 (mmcv); 1: float64 tlwh (deep_sort); 2: float32 y1x1y2x2 (TensorFlow).
    Returns kept indices (int64)."""
    dt = np.float64 if convention == 1 else np.float32
    boxes = np.ascontiguousarray(boxes, dt).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, dt).reshape(-1)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    k = C.c_int32()
    L.check(ctx.lib.pp_nms(ctx.handle, L.ptr(boxes), L.ptr(scores), n, float(thr), convention, L.ptr(keep), C.byref(k),
                           L.PP_MEM_HOST), "pp_nms")
    return keep[: k.value].astype(np.int64)


class TopDown:
    """pp_topdown handle: crop/normalise -> backbone -> flip-merge + decode, parameters resident."""

    def __init__(self, net, num_joints=17, flip_perm=None, shift_heatmap=True, post="unbiased", blur_kernel=17,
                 lut=None, chan_map=(0, 1, 2), in_name="input", out_name="output"):
        self.net = net
        self.ctx = net.ctx
        self.k = int(num_joints)
        lut = normalize_lut() if lut is None else np.ascontiguousarray(lut, np.float32)
        cm = np.asarray(chan_map, np.int32)
        perm = None if flip_perm is None else np.ascontiguousarray(flip_perm, np.int32)
        post_i = {"unbiased": 1, "default": 0, "udp": 2, None: -1}[post]   # "udp": UDP crop + DARK-UDP decode (ViTPose)
        h = C.c_void_p()
        L.check(self.ctx.lib.pp_topdown_create(net.handle, net.prog.named[in_name], net.prog.named[out_name], self.k,
                                               L.ptr(perm), int(shift_heatmap), post_i, int(blur_kernel), L.ptr(lut),
                                               L.ptr(cm), C.byref(h)), "pp_topdown_create")
        self.handle = h
        self.flip = perm is not None

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None) and getattr(self.net, "handle", None):
                self.ctx.lib.pp_topdown_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, frames, frame_idx, bboxes, frames_dev_shape=None):
        """frames: numpy [F][H][W][3] u8, or a device pointer (int) with frames_dev_shape=(F,H,W).
        Returns (kpts [P][K][3] fp32, valid [P] int32)."""
        bboxes = np.ascontiguousarray(bboxes, np.float64).reshape(-1, 4)
        p = bboxes.shape[0]
        frame_idx = np.ascontiguousarray(frame_idx, np.int32)
        if isinstance(frames, np.ndarray):
            frames = np.ascontiguousarray(frames, np.uint8)
            f, h, w, _ = frames.shape
            fmem = L.PP_MEM_HOST
        else:
            f, h, w = frames_dev_shape
            fmem = L.PP_MEM_DEVICE
        kp = np.empty((p, self.k, 3), np.float32)
        valid = np.zeros((p,), np.int32)
        L.check(self.ctx.lib.pp_topdown_run(self.handle, L.ptr(frames), f, h, w, fmem, L.ptr(frame_idx), L.ptr(bboxes), p,
                                            L.ptr(kp), L.PP_MEM_HOST, L.ptr(valid)), "pp_topdown_run")
        return kp, valid

    def run_precropped(self, x, center_scale, n=None):
        """x: numpy [N][H][W][4] fp32 or device pointer (int, with n given).  Returns kpts [N][K][3]."""
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, np.float32)
            n = x.shape[0]
            xmem = L.PP_MEM_HOST
        else:
            xmem = L.PP_MEM_DEVICE
        cs = np.ascontiguousarray(center_scale, np.float32).reshape(n, 4)
        kp = np.empty((n, self.k, 3), np.float32)
        L.check(self.ctx.lib.pp_topdown_run_precropped(self.handle, L.ptr(x), xmem, L.ptr(cs), n, L.ptr(kp),
                                                       L.PP_MEM_HOST), "pp_topdown_run_precropped")
        return kp

    def timing(self):
        """HIP-event stage times of the last run, ms: (pre, backbone, decode)."""
        ms = np.zeros(3, np.float32)
        L.check(self.ctx.lib.pp_topdown_timing(self.handle, L.ptr(ms)), "pp_topdown_timing")
        return tuple(float(v) for v in ms)
