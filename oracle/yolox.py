"""CPU oracle for the YOLOX detector of the ByteTrack configuration.  TEST INFRASTRUCTURE ONLY.

Selected by pose_pipeline/wrappers/mmtrack.py:8-29 with method "bytetrack" ->
3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private-half.py:9-20 on top of
3rdparty/mmtracking/_base_/models/yolox_x_8x8.py:4-27: CSPDarknet(deepen 1.33, widen 1.25) + YOLOXPAFPN(in [320, 640,
1280], out 320, 4 CSP blocks) + YOLOXHead(1 class, 320 channels), input scale (800, 1440), score_thr 0.01, NMS IoU 0.7;
test pipeline (same file, :62-78): Resize(keep_ratio) -> Normalize(mean 0, std 1, to_rgb False) -> Pad(32, value 114).
mmdet / mmcv are not vendored: the module internals (Focus, CSPLayer, DarknetBottleneck, SPPBottleneck, PAFPN wiring,
YOLOXHead decode, BN eps 1e-3, Swish) are restated from mmdet 2.x and are PARITY UNPINNED.  Parameter names are mmdet's
state_dict keys (prefix "detector."), so a real torch checkpoint would load.
"""
from __future__ import annotations

import numpy as np

from . import clib
from .boxes import nms_mmcv
from .detector import rescale_size, resize_linear_u8

F32 = np.float32
BN_EPS = 1e-3
STRIDES = (8, 16, 32)


def swish(x):
    s = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)          # torch.sigmoid
    return (x * s).astype(F32)


def preprocess(frame_rgb, scale=(800, 1440), divisor=32, pad_val=114.0):
    """-> (input [1][hp][wp][3] float32, scale_factor float32 [4] = (w, h, w, h))"""
    h, w = frame_rgb.shape[:2]
    nw, nh = rescale_size(w, h, scale)
    img = resize_linear_u8(frame_rgb, (nw, nh)).astype(F32)                 # Normalize(mean 0, std 1) is the identity
    hp, wp = -(-nh // divisor) * divisor, -(-nw // divisor) * divisor
    out = np.full((1, hp, wp, 3), pad_val, F32)
    out[0, :nh, :nw] = img
    sf = np.array([nw / w, nh / h, nw / w, nh / h], F32)
    return out, sf


class YOLOXRef:
    def __init__(self, sd, prefix="detector."):
        self.sd, self.p = sd, prefix

    def cm(self, x, name, stride=1, act=True):
        """mmcv ConvModule: conv (no bias) + BN(eps 1e-3) + Swish"""
        sd, p = self.sd, self.p + name
        w = sd[p + ".conv.weight"]
        scale = sd[p + ".bn.weight"].astype(np.float64) / np.sqrt(sd[p + ".bn.running_var"].astype(np.float64) + BN_EPS)
        wf = (w.astype(np.float64) * scale.reshape(-1, 1, 1, 1)).astype(F32)
        bf = (sd[p + ".bn.bias"].astype(np.float64) - sd[p + ".bn.running_mean"].astype(np.float64) * scale).astype(F32)
        k = w.shape[2]
        y = clib.conv2d_nhwc(x, wf, bf, stride=stride, pad=(k // 2, k // 2))
        return swish(y) if act else y

    def conv(self, x, name):
        sd, p = self.sd, self.p + name
        return clib.conv2d_nhwc(x, sd[p + ".weight"], sd[p + ".bias"])

    def csp(self, x, name, blocks, add_identity):
        short = self.cm(x, name + ".short_conv")
        main = self.cm(x, name + ".main_conv")
        for b in range(blocks):
            y = self.cm(self.cm(main, f"{name}.blocks.{b}.conv1"), f"{name}.blocks.{b}.conv2")
            main = (y + main).astype(F32) if add_identity else y
        return self.cm(np.concatenate([main, short], -1), name + ".final_conv")

    def backbone(self, x):
        # Focus: (top-left, bottom-left, top-right, bottom-right) patches stacked on the channel axis
        x = np.concatenate([x[:, ::2, ::2], x[:, 1::2, ::2], x[:, ::2, 1::2], x[:, 1::2, 1::2]], -1)
        x = self.cm(x, "backbone.stem.conv")
        outs = []
        for i, (blocks, ident, spp) in enumerate(((4, True, False), (12, True, False), (12, True, False), (4, False, True))):
            s = f"backbone.stage{i + 1}"
            x = self.cm(x, s + ".0", stride=2)
            j = 1
            if spp:
                y = self.cm(x, s + ".1.conv1")
                y = np.concatenate([y] + [clib.maxpool2d_nhwc(y, k, 1, k // 2) for k in (5, 9, 13)], -1)
                x = self.cm(y, s + ".1.conv2")
                j = 2
            x = self.csp(x, f"{s}.{j}", blocks, ident)
            if i >= 1:
                outs.append(x)
        return outs

    def neck(self, feats, blocks=4):
        up2 = lambda t: np.repeat(np.repeat(t, 2, 1), 2, 2)
        inner = [feats[-1]]
        for idx in (2, 1):
            high = self.cm(inner[0], f"neck.reduce_layers.{2 - idx}")
            inner[0] = high
            inner.insert(0, self.csp(np.concatenate([up2(high), feats[idx - 1]], -1), f"neck.top_down_blocks.{2 - idx}", blocks, False))
        outs = [inner[0]]
        for idx in (0, 1):
            down = self.cm(outs[-1], f"neck.downsamples.{idx}", stride=2)
            outs.append(self.csp(np.concatenate([down, inner[idx + 1]], -1), f"neck.bottom_up_blocks.{idx}", blocks, False))
        return [self.cm(o, f"neck.out_convs.{i}") for i, o in enumerate(outs)]

    def head(self, feats):
        cls, reg, obj = [], [], []
        for l, x in enumerate(feats):
            c = r = x
            for j in range(2):
                c = self.cm(c, f"bbox_head.multi_level_cls_convs.{l}.{j}")
                r = self.cm(r, f"bbox_head.multi_level_reg_convs.{l}.{j}")
            cls.append(self.conv(c, f"bbox_head.multi_level_conv_cls.{l}"))
            reg.append(self.conv(r, f"bbox_head.multi_level_conv_reg.{l}"))
            obj.append(self.conv(r, f"bbox_head.multi_level_conv_obj.{l}"))
        return cls, reg, obj

    def forward(self, x):
        return self.head(self.neck(self.backbone(x)))


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def detections(cls, reg, obj, scale_factor, score_thr=0.01, iou_thr=0.7):
    """YOLOXHead.get_bboxes for one image -> [n][5] float32 (x1, y1, x2, y2, score), descending score"""
    priors, cs, rs, os_ = [], [], [], []
    for s, c, r, o in zip(STRIDES, cls, reg, obj):
        h, w = c.shape[1:3]
        xx = np.tile(np.arange(w, dtype=F32) * F32(s), h)
        yy = np.repeat(np.arange(h, dtype=F32) * F32(s), w)
        priors.append(np.stack([xx, yy, np.full_like(xx, s), np.full_like(xx, s)], -1))
        cs.append(c.reshape(-1, c.shape[-1]))
        rs.append(r.reshape(-1, 4))
        os_.append(o.reshape(-1))
    priors, c, r, o = np.concatenate(priors), sigmoid(np.concatenate(cs)), np.concatenate(rs), sigmoid(np.concatenate(os_))
    xys = (r[:, :2] * priors[:, 2:] + priors[:, :2]).astype(F32)
    whs = (np.exp(r[:, 2:].astype(np.float64)).astype(F32) * priors[:, 2:]).astype(F32)
    boxes = np.stack([xys[:, 0] - whs[:, 0] / F32(2), xys[:, 1] - whs[:, 1] / F32(2),
                      xys[:, 0] + whs[:, 0] / F32(2), xys[:, 1] + whs[:, 1] / F32(2)], -1).astype(F32)
    boxes = (boxes / scale_factor[None]).astype(F32)
    max_scores = c.max(1)
    valid = (o * max_scores) >= F32(score_thr)
    boxes, scores = boxes[valid], (max_scores[valid] * o[valid]).astype(F32)
    if len(scores) == 0:
        return np.zeros((0, 5), F32)
    keep = nms_mmcv(boxes, scores, iou_thr)
    return np.concatenate([boxes[keep], scores[keep, None]], 1).astype(F32)


def detect(model: YOLOXRef, frame_rgb, scale=(800, 1440)):
    x, sf = preprocess(frame_rgb, scale)
    return detections(*model.forward(x), sf)
