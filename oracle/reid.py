"""CPU oracle for the appearance encoder of the DeepSortYOLOv4 tracking method.  TEST INFRASTRUCTURE ONLY.

Restates pose_pipeline/wrappers/deep_sort_yolov4/tools/:
  * generate_detections.py:25-63   extract_image_patch (aspect correction, int truncation, clipping, cv2.resize)
  * generate_detections.py:92-105  create_box_encoder (failed patch -> random patch; not reproduced: callers get None)
  * freeze_model.py:119-229        the mars-small128 network: conv1_1, conv1_2 (3x3, BN, ELU), max_pool 3x3/2 VALID,
                                   six pre-activation residual blocks (32, 32, 64, 64, 128, 128 channels), fc1 128 (BN, ELU),
                                   'ball' batch norm, L2 normalisation;  :239-240 BGR -> RGB inside the graph
extract_image_patch's box arithmetic is PINNED by tests/golden/reid_patch.npz (the reference function run here with
cv2.resize stubbed out).  TensorFlow is not installed and mars-small128.pb is absent: the network is PARITY UNPINNED.
slim defaults that matter: batch_norm(epsilon=1e-3, scale=False: no gamma), conv2d padding SAME (TensorFlow puts the odd
padding row / column at the END), dropout is the identity at inference.  Standalone batch norms (the pre-activation of
a residual link, 'ball') are y = fl(fl(x * s) + b) with s, b folded in float64 -- exactly a 1x1 diagonal convolution.
cv2.resize INTER_LINEAR on u8 is the fixed-point routine restated in oracle/detector.py.
"""
from __future__ import annotations

import numpy as np

from . import clib
from .detector import resize_linear_u8

F32 = np.float32
EPS = 1e-3


def patch_rect(bbox_tlwh, image_hw, patch_hw=(128, 64)):
    """generate_detections.py:44-60 -> (sx, sy, ex, ey) or None"""
    bbox = np.array(bbox_tlwh)                      # keeps the caller's dtype, like the reference
    target_aspect = float(patch_hw[1]) / patch_hw[0]
    new_width = target_aspect * bbox[3]
    bbox[0] -= (new_width - bbox[2]) / 2
    bbox[2] = new_width
    bbox[2:] += bbox[:2]
    bbox = bbox.astype(int)
    bbox[:2] = np.maximum(0, bbox[:2])
    bbox[2:] = np.minimum(np.asarray(image_hw[::-1]) - 1, bbox[2:])
    if np.any(bbox[:2] >= bbox[2:]):
        return None
    return tuple(int(v) for v in bbox)


def extract_image_patch(image_bgr, bbox_tlwh, patch_hw=(128, 64)):
    r = patch_rect(bbox_tlwh, image_bgr.shape[:2], patch_hw)
    if r is None:
        return None
    sx, sy, ex, ey = r
    return resize_linear_u8(np.ascontiguousarray(image_bgr[sy:ey, sx:ex]), (patch_hw[1], patch_hw[0]))


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0).astype(np.float64))).astype(F32)


def bn_affine(p, prefix):
    """slim.batch_norm inference as (scale, shift), float64 -> float32; gamma is optional (scale=False)."""
    var = p[prefix + ".var"].astype(np.float64)
    g = p.get(prefix + ".gamma")
    s = (1.0 if g is None else g.astype(np.float64)) / np.sqrt(var + EPS)
    b = p[prefix + ".beta"].astype(np.float64) - p[prefix + ".mean"].astype(np.float64) * s
    return s.astype(F32), b.astype(F32)


def same_pad(x, k, stride):
    """TensorFlow SAME: total = max((ceil(n/s)-1)*s + k - n, 0), the extra unit goes last."""
    pads = []
    for n in x.shape[1:3]:
        total = max((-(-n // stride) - 1) * stride + k - n, 0)
        pads.append((total // 2, total - total // 2))
    return np.pad(x, ((0, 0), pads[0], pads[1], (0, 0)))


class MarsSmall128Ref:
    """state dict (torch conv layout [cout][cin][kh][kw]): conv1_1, conv1_2, conv{2,3,4}_{1,3}/{1,2}, .../projection,
    fc1.weight [128][16*8*128 in (h, w, c) order], *.bias, *.bn.{beta,mean,var}, ball.{beta,mean,var}."""

    def __init__(self, sd):
        self.p = sd

    def conv(self, x, name, stride=1, bn=True, act=True):
        w = self.p[name + ".weight"]
        # slim creates no bias when a normalizer_fn is given (conv + BN layers, fc1) or biases_initializer=None (projection)
        b = self.p[name + ".bias"].astype(np.float64) if name + ".bias" in self.p else np.zeros(w.shape[0])
        if bn:
            # BN(conv + bias): fold in float64, one rounding (same formula as the product's fold)
            var = self.p[name + ".bn.var"].astype(np.float64)
            sc = 1.0 / np.sqrt(var + EPS)
            wf = (w.astype(np.float64) * sc.reshape(-1, 1, 1, 1)).astype(F32)
            bf = (self.p[name + ".bn.beta"].astype(np.float64) + (b - self.p[name + ".bn.mean"].astype(np.float64)) * sc).astype(F32)
        else:
            wf, bf = w, b.astype(F32)
        y = clib.conv2d_nhwc(same_pad(x, w.shape[2], stride), wf, bf, stride=stride, pad=(0, 0))
        return elu(y) if act else y

    def block(self, x, scope, increase_dim=False, is_first=False):
        if is_first:
            net = x
        else:
            s, b = bn_affine(self.p, scope + ".bn")
            net = elu((x * s).astype(F32) + b)
        stride = 2 if increase_dim else 1
        y = self.conv(net, scope + ".1", stride=stride)
        y = self.conv(y, scope + ".2", bn=False, act=False)
        if increase_dim:
            x = self.conv(x, scope + ".projection", stride=2, bn=False, act=False)
        return (x + y).astype(F32)

    def forward(self, patches_bgr_u8):
        """[N][128][64][3] u8 BGR -> [N][128] float32 unit-norm features"""
        x = patches_bgr_u8[..., ::-1].astype(F32)                     # freeze_model.py:239-240, :255
        x = self.conv(x, "conv1_1")
        x = self.conv(x, "conv1_2")
        x = clib.maxpool2d_nhwc(x, 3, 2, 0)                           # VALID
        x = self.block(x, "conv2_1", is_first=True)
        x = self.block(x, "conv2_3")
        x = self.block(x, "conv3_1", increase_dim=True)
        x = self.block(x, "conv3_3")
        x = self.block(x, "conv4_1", increase_dim=True)
        x = self.block(x, "conv4_3")
        n, h, w, c = x.shape
        wfc = self.p["fc1.weight"].reshape(-1, h, w, c).transpose(0, 3, 1, 2)       # [128][(h, w, c)] -> conv layout
        var = self.p["fc1.bn.var"].astype(np.float64)
        sc = 1.0 / np.sqrt(var + EPS)
        wf = (wfc.astype(np.float64) * sc.reshape(-1, 1, 1, 1)).astype(F32)
        b0 = self.p["fc1.bias"].astype(np.float64) if "fc1.bias" in self.p else 0.0
        bf = (self.p["fc1.bn.beta"].astype(np.float64) + (b0 - self.p["fc1.bn.mean"].astype(np.float64)) * sc).astype(F32)
        f = elu(clib.conv2d_nhwc(x, np.ascontiguousarray(wf), bf, stride=1, pad=(0, 0))).reshape(n, -1)
        s, b = bn_affine(self.p, "ball")
        f = ((f * s).astype(F32) + b).astype(F32)
        out = np.empty_like(f)
        for i in range(n):
            acc = F32(0)
            for v in f[i]:
                acc = F32(acc + F32(v * v))
            norm = F32(np.sqrt(F32(F32(1e-8) + acc)))
            out[i] = f[i] / norm
        return out

    def encode(self, frame_bgr, boxes_tlwh):
        """create_box_encoder(frame, boxes): float64 [n][128] (generate_detections.py:80)"""
        patches = [extract_image_patch(frame_bgr, b) for b in boxes_tlwh]
        # empty patch: the reference encodes an unseeded RANDOM patch (:98-101); the build encodes zeros instead
        patches = [np.zeros((128, 64, 3), np.uint8) if p is None else p for p in patches]
        if not patches:
            return np.zeros((0, 128), float)
        return self.forward(np.stack(patches)).astype(float)
