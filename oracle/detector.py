"""CPU oracle for the detection stage.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `mmtrack.apis.inference_mot(model, frame, frame_id)` (pose_pipeline/wrappers/mmtrack.py:45)
computes before association, for the detector wired by
3rdparty/mmtracking/mot/deepsort/sort_faster-rcnn_fpn_4e_mot17-private-half.py:5-15 on top of
3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112 and the test pipeline of
3rdparty/mmtracking/_base_/datasets/mot_challenge.py:3-4,33-47:
   Resize(keep_ratio, (1088,1088)) -> Normalize(to_rgb) -> Pad(/32) -> ResNet-50 -> FPN -> RPNHead ->
   proposals (top-1000/level, delta2bbox, batched NMS .7, top 1000) -> SingleRoIExtractor / RoIAlign 7x7 ->
   Shared2FCBBoxHead (1 class) -> softmax, delta2bbox(.1,.1,.2,.2), rescale, score>.05, NMS .5, top 100.
mmdet 2.x / mmcv-full 1.x / OpenCV are un-vendored, un-pinned third-party dependencies
(requirements.txt:9-12 are comments) and are not installed here: PARITY UNPINNED against the real
packages.  Published algorithms restated: mmcv `imrescale` + cv2.resize(INTER_LINEAR) 8-bit fixed point,
mmcv `imnormalize`, mmdet ResNet(style='pytorch') / FPN / AnchorGenerator / DeltaXYWHBBoxCoder /
RPNHead._get_bboxes_single / batched_nms / SingleRoIExtractor.map_roi_levels / mmcv RoIAlign
(aligned=True, sampling_ratio=0, avg) / multiclass_nms.
Transcendentals (sigmoid, exp, softmax) are evaluated in float64 and rounded once to float32 so that the
value does not depend on a libm; ties in score sorts break towards the lower index.
"""
from __future__ import annotations

import numpy as np

from . import boxes as obox
from . import clib
from .nets import fold_bn, relu

f32 = np.float32
MEAN = np.array([123.675, 116.28, 103.53], dtype=np.float64)
STD = np.array([58.395, 57.12, 57.375], dtype=np.float64)


# ---- pre-processing ---------------------------------------------------------------------------------
def rescale_size(w, h, scale=(1088, 1088)):
    """mmcv.rescale_size: keep ratio so that the image fits `scale`."""
    max_long, max_short = max(scale), min(scale)
    sf = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)


def _resize_coeffs(src, dst):
    """cv::resize INTER_LINEAR index/weight tables for one axis (8-bit path, weights * 2048)."""
    scale = 1.0 / (dst / src)
    idx = np.zeros(dst, np.int64)
    w = np.zeros((dst, 2), np.int64)
    for d in range(dst):
        fx = f32((d + 0.5) * scale - 0.5)
        s = int(np.floor(fx))
        fx = f32(fx - f32(s))
        if s < 0:
            fx, s = f32(0), 0
        if s >= src - 1:
            fx, s = f32(0), src - 1
        idx[d] = s
        w[d, 0] = int(np.rint(f32(f32(1.0) - fx) * f32(2048)))
        w[d, 1] = int(np.rint(fx * f32(2048)))
    return idx, w


def resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize, interpolation=cv2.INTER_LINEAR) for HxWxC uint8."""
    h, w, _ = img.shape
    dw, dh = dsize
    xi, xw = _resize_coeffs(w, dw)
    yi, yw = _resize_coeffs(h, dh)
    src = img.astype(np.int64)
    x1 = np.minimum(xi + 1, w - 1)
    hor = src[:, xi] * xw[None, :, 0, None] + src[:, x1] * xw[None, :, 1, None]          # [h][dw][c], *2048
    y1 = np.minimum(yi + 1, h - 1)
    r0, r1 = hor[yi], hor[y1]
    out = (((yw[:, 0, None, None] * (r0 >> 4)) >> 16) + ((yw[:, 1, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def normalize_lut():
    """mmcv.imnormalize on a float32 image: (v - mean) * (1/std) with the scalars cast to float32."""
    v = np.arange(256, dtype=f32)
    return (((v[None, :] - MEAN.astype(f32)[:, None]).astype(f32)) * (1.0 / STD).astype(f32)[:, None]).astype(f32)


def preprocess(frame_wrapper_rgb, scale=(1088, 1088), divisor=32):
    """frame_wrapper_rgb: the array the wrapper hands to mmtrack (cv2 BGR frame after the wrapper's
    BGR2RGB, wrappers/mmtrack.py:43).  mmcv's Normalize(to_rgb=True) swaps channels again, so tensor
    channel 0 is the ORIGINAL B plane normalised with mean 123.675.
    Returns (tensor [Hp][Wp][3] float32 NHWC, scale_factor float32[4], (new_h, new_w))."""
    h, w, _ = frame_wrapper_rgb.shape
    nw, nh = rescale_size(w, h, scale)
    img = resize_linear_u8(frame_wrapper_rgb, (nw, nh))
    img = img[:, :, ::-1]                                           # to_rgb
    lut = normalize_lut()
    t = np.stack([lut[c][img[:, :, c]] for c in range(3)], axis=-1)
    hp, wp = (nh + divisor - 1) // divisor * divisor, (nw + divisor - 1) // divisor * divisor
    out = np.zeros((hp, wp, 3), f32)
    out[:nh, :nw] = t
    sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=f32)
    return out, sf, (nh, nw)


# ---- backbone + neck + RPN head ---------------------------------------------------------------------
class FasterRCNNRef:
    def __init__(self, sd, prefix="detector."):
        self.sd, self.p = sd, prefix

    def _cb(self, x, conv, bn, stride=1, pad=0):
        sd, p = self.sd, self.p
        w, b = fold_bn(sd[p + conv + ".weight"], sd[p + bn + ".weight"], sd[p + bn + ".bias"], sd[p + bn + ".running_mean"],
                       sd[p + bn + ".running_var"])
        return clib.conv2d_nhwc(x, w, b, stride=stride, pad=pad)

    def _conv(self, x, name, stride=1, pad=0):
        sd, p = self.sd, self.p
        return clib.conv2d_nhwc(x, sd[p + name + ".weight"], sd[p + name + ".bias"], stride=stride, pad=pad)

    def backbone(self, x):
        x = relu(self._cb(x, "backbone.conv1", "backbone.bn1", 2, 3))
        x = clib.maxpool2d_nhwc(x, 3, 2, 1)
        outs = []
        for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2))):
            for b in range(blocks):
                q = f"backbone.layer{li + 1}.{b}."
                s = stride if b == 0 else 1
                idn = self._cb(x, q + "downsample.0", q + "downsample.1", s, 0) if b == 0 else x
                y = relu(self._cb(x, q + "conv1", q + "bn1", 1, 0))
                y = relu(self._cb(y, q + "conv2", q + "bn2", s, 1))       # style='pytorch': stride on the 3x3
                y = self._cb(y, q + "conv3", q + "bn3", 1, 0)
                x = relu(y + idn)
            outs.append(x)
        return outs

    def fpn(self, feats):
        lat = [self._conv(f, f"neck.lateral_convs.{i}.conv") for i, f in enumerate(feats)]
        for i in range(3, 0, -1):
            up = np.repeat(np.repeat(lat[i], 2, axis=1), 2, axis=2)[:, : lat[i - 1].shape[1], : lat[i - 1].shape[2]]
            lat[i - 1] = lat[i - 1] + up
        outs = [self._conv(l, f"neck.fpn_convs.{i}.conv", 1, 1) for i, l in enumerate(lat)]
        outs.append(outs[-1][:, ::2, ::2].copy())                       # F.max_pool2d(x, 1, stride=2)
        return outs

    def rpn_head(self, feats):
        cls, reg = [], []
        for f in feats:
            t = relu(self._conv(f, "rpn_head.rpn_conv", 1, 1))
            cls.append(self._conv(t, "rpn_head.rpn_cls"))
            reg.append(self._conv(t, "rpn_head.rpn_reg"))
        return cls, reg

    def roi_head(self, roi_feats):
        """roi_feats [R][7][7][256] NHWC -> (cls [R][2], reg [R][4]).  The flatten order of mmdet is
        (c, h, w): fc weight [1024][12544] == conv weight [1024][256][7][7]."""
        sd, p = self.sd, self.p
        w0 = sd[p + "roi_head.bbox_head.shared_fcs.0.weight"].reshape(1024, 256, 7, 7)
        x = relu(clib.conv2d_nhwc(roi_feats, w0, sd[p + "roi_head.bbox_head.shared_fcs.0.bias"]))
        w1 = sd[p + "roi_head.bbox_head.shared_fcs.1.weight"][:, :, None, None]
        x = relu(clib.conv2d_nhwc(x, w1, sd[p + "roi_head.bbox_head.shared_fcs.1.bias"]))
        cls = clib.conv2d_nhwc(x, sd[p + "roi_head.bbox_head.fc_cls.weight"][:, :, None, None], sd[p + "roi_head.bbox_head.fc_cls.bias"])
        reg = clib.conv2d_nhwc(x, sd[p + "roi_head.bbox_head.fc_reg.weight"][:, :, None, None], sd[p + "roi_head.bbox_head.fc_reg.bias"])
        return cls.reshape(-1, cls.shape[-1]), reg.reshape(-1, 4)


# ---- anchors, box coding, proposals -----------------------------------------------------------------
STRIDES = (4, 8, 16, 32, 64)


def base_anchors(stride, scales=(8,), ratios=(0.5, 1.0, 2.0)):
    """AnchorGenerator.gen_single_level_base_anchors (center_offset 0, scale_major) in float32."""
    r = np.array(ratios, f32)
    s = np.array(scales, f32)
    h_ratios = np.sqrt(r).astype(f32)
    w_ratios = (f32(1) / h_ratios).astype(f32)
    ws = ((f32(stride) * w_ratios[:, None]).astype(f32) * s[None, :]).astype(f32).reshape(-1)
    hs = ((f32(stride) * h_ratios[:, None]).astype(f32) * s[None, :]).astype(f32).reshape(-1)
    return np.stack([f32(-0.5) * ws, f32(-0.5) * hs, f32(0.5) * ws, f32(0.5) * hs], axis=-1).astype(f32)


def grid_anchors(feat_h, feat_w, stride):
    base = base_anchors(stride)
    sx = (np.arange(feat_w, dtype=f32) * f32(stride)).astype(f32)
    sy = (np.arange(feat_h, dtype=f32) * f32(stride)).astype(f32)
    xx = np.tile(sx, feat_h)
    yy = np.repeat(sy, feat_w)
    shifts = np.stack([xx, yy, xx, yy], axis=-1)
    return (base[None, :, :] + shifts[:, None, :]).astype(f32).reshape(-1, 4)


MAX_RATIO = float(np.abs(np.log(16 / 1000)))


def exp_f32(x):
    return np.exp(x.astype(np.float64)).astype(f32)


def sigmoid_f32(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def delta2bbox(rois, deltas, stds=(1.0, 1.0, 1.0, 1.0)):
    """DeltaXYWHBBoxCoder.decode with means 0, clip_border=False, float32."""
    d = (deltas.astype(f32) * np.array(stds, f32)[None, :]).astype(f32)
    pxy = ((rois[:, :2] + rois[:, 2:]).astype(f32) * f32(0.5)).astype(f32)
    pwh = (rois[:, 2:] - rois[:, :2]).astype(f32)
    dxy_wh = (pwh * d[:, :2]).astype(f32)
    dwh = np.clip(d[:, 2:], f32(-MAX_RATIO), f32(MAX_RATIO)).astype(f32)
    gxy = (pxy + dxy_wh).astype(f32)
    gwh = (pwh * exp_f32(dwh)).astype(f32)
    half = (gwh * f32(0.5)).astype(f32)
    return np.concatenate([(gxy - half).astype(f32), (gxy + half).astype(f32)], axis=-1)


def rpn_proposals(cls_maps, reg_maps, nms_pre=1000, max_per_img=1000, iou_thr=0.7):
    """RPNHead._get_bboxes_single + _bbox_post_process for one image.
    cls_maps[l] [H][W][3] logits, reg_maps[l] [H][W][12].  Returns (proposals [P][4], scores [P])."""
    scores_l, boxes_l, ids_l = [], [], []
    for lvl, (c, r) in enumerate(zip(cls_maps, reg_maps)):
        h, w, _ = c.shape
        scores = sigmoid_f32(c.reshape(-1))
        deltas = r.reshape(-1, 4)
        anchors = grid_anchors(h, w, STRIDES[lvl])
        if 0 < nms_pre < scores.shape[0]:
            order = np.argsort(-scores, kind="stable")[:nms_pre]
            scores, deltas, anchors = scores[order], deltas[order], anchors[order]
        scores_l.append(scores)
        boxes_l.append(delta2bbox(anchors, deltas))
        ids_l.append(np.full(scores.shape[0], lvl, np.int64))
    scores = np.concatenate(scores_l)
    props = np.concatenate(boxes_l)
    ids = np.concatenate(ids_l)
    valid = ((props[:, 2] - props[:, 0]) > 0) & ((props[:, 3] - props[:, 1]) > 0)      # min_bbox_size = 0
    props, scores, ids = props[valid], scores[valid], ids[valid]
    if props.shape[0] == 0:
        return np.zeros((0, 4), f32), np.zeros((0,), f32)
    keep = batched_nms(props, scores, ids, iou_thr)[:max_per_img]
    return props[keep], scores[keep]


def batched_nms(boxes, scores, ids, iou_thr):
    """mmcv.ops.batched_nms: per-class NMS through coordinate offsets (float32)."""
    max_coordinate = boxes.max()
    offsets = (ids.astype(f32) * f32(max_coordinate + f32(1))).astype(f32)
    return np.array(obox.nms_mmcv((boxes + offsets[:, None]).astype(f32), scores, iou_thr), dtype=np.int64)


# ---- RoI feature extraction ---------------------------------------------------------------------------
def map_roi_levels(rois, num_levels=4, finest_scale=56):
    scale = np.sqrt(((rois[:, 2] - rois[:, 0]).astype(f32) * (rois[:, 3] - rois[:, 1]).astype(f32)).astype(f32)).astype(f32)
    lvls = np.floor(np.log2((scale / f32(finest_scale)).astype(f32) + f32(1e-6)))
    return np.clip(lvls, 0, num_levels - 1).astype(np.int64)


def _bilinear(feat, y, x):
    h, w, _ = feat.shape
    if y < -1.0 or y > h or x < -1.0 or x > w:
        return np.zeros(feat.shape[2], f32)
    y = f32(max(y, f32(0)))
    x = f32(max(x, f32(0)))
    y_low, x_low = int(y), int(x)
    if y_low >= h - 1:
        y_high = y_low = h - 1
        y = f32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= w - 1:
        x_high = x_low = w - 1
        x = f32(x_low)
    else:
        x_high = x_low + 1
    ly, lx = f32(y - f32(y_low)), f32(x - f32(x_low))
    hy, hx = f32(f32(1) - ly), f32(f32(1) - lx)
    w1, w2, w3, w4 = f32(hy * hx), f32(hy * lx), f32(ly * hx), f32(ly * lx)
    v = (w1 * feat[y_low, x_low]).astype(f32)
    v = (v + (w2 * feat[y_low, x_high]).astype(f32)).astype(f32)
    v = (v + (w3 * feat[y_high, x_low]).astype(f32)).astype(f32)
    return (v + (w4 * feat[y_high, x_high]).astype(f32)).astype(f32)


def roi_align(feat, roi, spatial_scale, out=7):
    """mmcv RoIAlign(output_size=7, sampling_ratio=0, pool_mode='avg', aligned=True) for one roi on an
    [H][W][C] map -> [7][7][C], float32 accumulation in (iy, ix) order."""
    ss = f32(spatial_scale)
    x1, y1 = f32(f32(roi[0] * ss) - f32(0.5)), f32(f32(roi[1] * ss) - f32(0.5))
    x2, y2 = f32(f32(roi[2] * ss) - f32(0.5)), f32(f32(roi[3] * ss) - f32(0.5))
    rw, rh = f32(x2 - x1), f32(y2 - y1)
    bw, bh = f32(rw / f32(out)), f32(rh / f32(out))
    gh = int(np.ceil(f32(rh / f32(out))))
    gw = int(np.ceil(f32(rw / f32(out))))
    count = f32(max(gh * gw, 1))
    res = np.zeros((out, out, feat.shape[2]), f32)
    for ph in range(out):
        for pw in range(out):
            acc = np.zeros(feat.shape[2], f32)
            for iy in range(gh):
                y = f32(f32(y1 + f32(f32(ph) * bh)) + f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh))
                for ix in range(gw):
                    x = f32(f32(x1 + f32(f32(pw) * bw)) + f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw))
                    acc = (acc + _bilinear(feat, y, x)).astype(f32)
            res[ph, pw] = (acc / count).astype(f32)
    return res


def extract_roi_feats(fpn_feats, rois):
    lv = map_roi_levels(rois)
    out = np.zeros((rois.shape[0], 7, 7, fpn_feats[0].shape[-1]), f32)
    for i, (r, l) in enumerate(zip(rois, lv)):
        out[i] = roi_align(fpn_feats[l][0], r, 1.0 / STRIDES[l])
    return out, lv


# ---- final detections ----------------------------------------------------------------------------------
def softmax_fg(cls):
    z = cls.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(f32)[:, 0]         # single class: column 0, background last


def final_detections(rois, cls, reg, scale_factor, score_thr=0.05, iou_thr=0.5, max_per_img=100):
    scores = softmax_fg(cls)
    boxes = delta2bbox(rois, reg, stds=(0.1, 0.1, 0.2, 0.2))
    boxes = (boxes / scale_factor[None, :].astype(f32)).astype(f32)
    inds = np.nonzero(scores > f32(score_thr))[0]
    boxes, scores = boxes[inds], scores[inds]
    if boxes.shape[0] == 0:
        return np.zeros((0, 5), f32)
    keep = batched_nms(boxes, scores, np.zeros(len(scores), np.int64), iou_thr)[:max_per_img]
    return np.concatenate([boxes[keep], scores[keep, None]], axis=1).astype(f32)


def detect(model: FasterRCNNRef, frame_wrapper_rgb, want_intermediates=False):
    """One frame -> [n][5] (x1, y1, x2, y2, score) in source pixels, like `result['det_bboxes'][0]`."""
    x, sf, _ = preprocess(frame_wrapper_rgb)
    feats = model.fpn(model.backbone(x[None]))
    cls_maps, reg_maps = model.rpn_head(feats)
    props, pscores = rpn_proposals([c[0] for c in cls_maps], [r[0] for r in reg_maps])
    roi_feats, lv = extract_roi_feats(feats[:4], props)
    cls, reg = model.roi_head(roi_feats)
    dets = final_detections(props, cls, reg, sf)
    if want_intermediates:
        return dets, dict(x=x, feats=feats, cls_maps=cls_maps, reg_maps=reg_maps, proposals=props, proposal_scores=pscores,
                          roi_levels=lv, roi_feats=roi_feats, cls=cls, reg=reg, scale_factor=sf)
    return dets
