"""CPU oracle for the YOLOv4 detector of the DeepSortYOLOv4 tracking method.  TEST INFRASTRUCTURE ONLY.

Restates pose_pipeline/wrappers/deep_sort_yolov4/ (tracking_method 0, the default of every recipe in
utils/standard_pipelines.py:12,58,112):
  * yolo4/utils.py:21-32   letterbox_image: PIL `Image.resize(BICUBIC)` + paste on a (128,128,128) canvas
  * yolo.py:85-99          detect_image pre-processing (float32 / 255)
  * yolo4/model.py:55-190  DarknetConv2D_BN_{Mish,Leaky}, resblock_body, darknet_body, yolo4_body
  * yolo4/model.py:193-293 yolo_head, yolo_correct_boxes, yolo_boxes_and_scores, yolo_eval (tf.image.non_max_suppression)
  * yolo.py:108-129        person filter, int() truncation and clipping of the boxes
PIL is importable here, so the bicubic letterbox is PINNED by tests/golden/letterbox.npz (generated from the
reference's own letterbox_image).  TensorFlow / Keras are not installed and the checkpoint (yolo4.h5) is absent:
the network and the float32 TensorFlow ops of the decode are PARITY UNPINNED -- they are restated op by op, every
transcendental evaluated in double precision and rounded to float32 once per reference op.
Keras defaults that matter: BatchNormalization(epsilon=1e-3); LeakyReLU(alpha=0.1); 'same' padding, except stride-2
convs which are 'valid' after ZeroPadding2D(((1,0),(1,0))) -- for the even sizes of a 416 input that equals a symmetric
pad of 1.  Layer parameters are named l0, l1, ... in the order yolo4_body creates its convolutions.
"""
from __future__ import annotations

import numpy as np

from . import clib

F32 = np.float32
# model_data/yolo_anchors.txt is not in the repository; these are the YOLOv4 anchors the keras-yolo4 project ships
ANCHORS = np.array([[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]], F32)
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]     # yolo4/model.py:259
PRECISION_BITS = 32 - 8 - 2                           # Pillow src/libImaging/Resample.c


# ---- PIL bicubic resize (Pillow ImagingResample, 8 bits per channel) -----------------------------------------
def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size, support=2.0, filt=_bicubic):
    """precompute_coeffs + normalize_coeffs_8bpc: per output index (xmin, count, int coefficients)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        ki = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in k]
        out.append((xmin, xmax, np.array(ki, np.int64)))
    return out


def _resample_axis1(img, out_size):
    """one ImagingResampleHorizontal_8bpc pass along axis 1 of a [H][W][C] u8 array"""
    co = pil_coeffs(img.shape[1], out_size)
    res = np.empty((img.shape[0], out_size, img.shape[2]), np.uint8)
    src = img.astype(np.int64)
    for xx, (xmin, cnt, k) in enumerate(co):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, xmin:xmin + cnt, :], k, axes=([1], [0]))
        res[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return res


def pil_resize_bicubic(img, size_wh):
    """Image.resize((w, h), Image.BICUBIC) of an RGB u8 array: horizontal pass, then vertical pass."""
    nw, nh = size_wh
    out = img
    if nw != img.shape[1]:
        out = _resample_axis1(out, nw)
    if nh != img.shape[0]:
        out = np.transpose(_resample_axis1(np.transpose(out, (1, 0, 2)), nh), (1, 0, 2))
    return np.ascontiguousarray(out)


def letterbox(img_rgb, size_wh=(416, 416)):
    """yolo4/utils.py:21-32 -> (boxed u8 image [h][w][3], (nw, nh, dx, dy))"""
    ih, iw = img_rgb.shape[:2]
    w, h = size_wh
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    small = pil_resize_bicubic(img_rgb, (nw, nh))
    canvas = np.full((h, w, 3), 128, np.uint8)
    dx, dy = (w - nw) // 2, (h - nh) // 2
    canvas[dy:dy + nh, dx:dx + nw] = small
    return canvas, (nw, nh, dx, dy)


def network_input(img_rgb, size_wh=(416, 416)):
    """yolo.py:93-99: float32 image / 255, batch axis added -> [1][h][w][3]"""
    boxed, _ = letterbox(img_rgb, size_wh)
    x = boxed.astype(F32)
    x /= F32(255.0)
    return x[None]


# ---- activations (one rounding per reference float32 op) ------------------------------------------------------
def leaky(x):
    return np.where(x >= 0, x, F32(0.1) * x).astype(F32)


def mish(x):
    """inputs * K.tanh(K.softplus(inputs)) (yolo4/model.py:48).  tanh(softplus(x)) is evaluated in float64 through the
    identity tanh(log(1 + e^x)) = n(n + 2) / (n(n + 2) + 2), n = e^x, and rounded to float32 once; the product is a
    float32 multiply."""
    n = np.exp(np.minimum(x, F32(20)).astype(np.float64))
    t = n * (n + 2.0)
    th = np.where(x > F32(20), 1.0, t / (t + 2.0)).astype(F32)
    return (x * th).astype(F32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def exp32(x):
    return np.exp(x.astype(np.float64)).astype(F32)


# ---- network ----------------------------------------------------------------------------------------------------
def fold_bn_keras(w, gamma, beta, mean, var, eps=1e-3):
    scale = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps)
    wf = (w.astype(np.float64) * scale.reshape(-1, 1, 1, 1)).astype(F32)
    bf = (beta.astype(np.float64) - mean.astype(np.float64) * scale).astype(F32)
    return wf, bf


class YOLOv4Ref:
    """state dict: l{i}.weight [cout][cin][kh][kw], l{i}.bn.{gamma,beta,mean,var} or l{i}.bias, i in creation order."""

    def __init__(self, sd, num_classes=80):
        self.sd = sd
        self.nc = num_classes
        self.i = 0

    def _conv(self, x, stride=1, act=None):
        i = self.i
        self.i += 1
        w = self.sd[f"l{i}.weight"]
        if f"l{i}.bias" in self.sd:
            wf, bf = w, self.sd[f"l{i}.bias"]
        else:
            wf, bf = fold_bn_keras(w, *(self.sd[f"l{i}.bn.{k}"] for k in ("gamma", "beta", "mean", "var")))
        k = w.shape[2]
        y = clib.conv2d_nhwc(x, wf, bf, stride=stride, pad=(k // 2, k // 2))
        return y if act is None else act(y)

    def _resblock(self, x, filters, blocks, all_narrow=True):
        half = filters // 2 if all_narrow else filters
        pre = self._conv(x, 2, mish)
        short = self._conv(pre, 1, mish)
        main = self._conv(pre, 1, mish)
        for _ in range(blocks):
            y = self._conv(self._conv(main, 1, mish), 1, mish)
            main = (main + y).astype(F32)
        post = self._conv(main, 1, mish)
        assert post.shape[-1] == half
        return self._conv(np.concatenate([post, short], -1), 1, mish)

    def forward(self, x):
        """x: [N][H][W][3] float32 -> [y19, y38, y76] raw head outputs [N][h][w][3*(5+nc)]"""
        self.i = 0
        c = self._conv
        x = c(x, 1, mish)
        x = self._resblock(x, 64, 1, False)
        x = self._resblock(x, 128, 2)
        f76 = x = self._resblock(x, 256, 8)            # darknet.layers[131]
        f38 = x = self._resblock(x, 512, 8)            # darknet.layers[204]
        x = self._resblock(x, 1024, 4)
        y19 = c(c(c(x, 1, leaky), 1, leaky), 1, leaky)
        mp = [clib.maxpool2d_nhwc(y19, k, 1, k // 2) for k in (13, 9, 5)]
        y19 = np.concatenate(mp + [y19], -1)
        y19 = c(c(c(y19, 1, leaky), 1, leaky), 1, leaky)
        up = np.repeat(np.repeat(c(y19, 1, leaky), 2, 1), 2, 2)
        y38 = np.concatenate([c(f38, 1, leaky), up], -1)
        for _ in range(5):
            y38 = c(y38, 1, leaky)
        up = np.repeat(np.repeat(c(y38, 1, leaky), 2, 1), 2, 2)
        y76 = np.concatenate([c(f76, 1, leaky), up], -1)
        for _ in range(5):
            y76 = c(y76, 1, leaky)
        y76_out = c(c(y76, 1, leaky), 1, None)
        y38 = np.concatenate([c(y76, 2, leaky), y38], -1)
        for _ in range(5):
            y38 = c(y38, 1, leaky)
        y38_out = c(c(y38, 1, leaky), 1, None)
        y19 = np.concatenate([c(y38, 2, leaky), y19], -1)
        for _ in range(5):
            y19 = c(y19, 1, leaky)
        y19_out = c(c(y19, 1, leaky), 1, None)
        return [y19_out, y38_out, y76_out]


# ---- decode (float32 TensorFlow ops, yolo4/model.py:193-254) --------------------------------------------------
def boxes_and_scores(feats, anchors, num_classes, input_hw, image_hw):
    n, gh, gw, _ = feats.shape
    assert n == 1
    f = feats.reshape(gh, gw, len(anchors), num_classes + 5)
    gx = np.tile(np.arange(gw, dtype=F32).reshape(1, gw, 1, 1), (gh, 1, 1, 1))
    gy = np.tile(np.arange(gh, dtype=F32).reshape(gh, 1, 1, 1), (1, gw, 1, 1))
    grid = np.concatenate([gx, gy], -1)
    box_xy = (sigmoid(f[..., :2]) + grid) / np.array([gw, gh], F32)
    box_wh = exp32(f[..., 2:4]) * anchors.reshape(1, 1, -1, 2).astype(F32) / np.array(input_hw[::-1], F32)
    conf = sigmoid(f[..., 4:5])
    prob = sigmoid(f[..., 5:])
    # yolo_correct_boxes
    box_yx, box_hw = box_xy[..., ::-1], box_wh[..., ::-1]
    inp = np.array(input_hw, F32)
    img = np.array(image_hw, F32)
    new_shape = np.round(img * np.min(inp / img)).astype(F32)       # K.round: half to even, like np.round
    offset = (inp - new_shape) / F32(2.0) / inp
    scale = inp / new_shape
    box_yx = (box_yx - offset) * scale
    box_hw = box_hw * scale
    mins = box_yx - box_hw / F32(2.0)
    maxes = box_yx + box_hw / F32(2.0)
    boxes = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], -1)
    boxes = boxes * np.concatenate([img, img])
    scores = conf * prob
    return boxes.reshape(-1, 4).astype(F32), scores.reshape(-1, num_classes).astype(F32)


def tf_nms(boxes, scores, max_out, iou_thr):
    """tf.image.non_max_suppression: descending score (ties: lower index first), suppress when IoU > thr."""
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    y1 = np.minimum(boxes[:, 0], boxes[:, 2]); y2 = np.maximum(boxes[:, 0], boxes[:, 2])
    x1 = np.minimum(boxes[:, 1], boxes[:, 3]); x2 = np.maximum(boxes[:, 1], boxes[:, 3])
    area = ((y2 - y1) * (x2 - x1)).astype(F32)
    keep = []
    for i in order:
        if len(keep) >= max_out:
            break
        ok = True
        for j in keep:
            if area[i] <= 0 or area[j] <= 0:
                continue
            ih = max(F32(min(y2[i], y2[j]) - max(y1[i], y1[j])), F32(0))
            iw = max(F32(min(x2[i], x2[j]) - max(x1[i], x1[j])), F32(0))
            inter = F32(ih * iw)
            if F32(inter / F32(F32(area[i] + area[j]) - inter)) > F32(iou_thr):
                ok = False
                break
        if ok:
            keep.append(i)
    return np.array(keep, np.int64)


def person_detections(outputs, image_hw, anchors=ANCHORS, num_classes=80, score_thr=0.5, iou_thr=0.5, max_boxes=200):
    """yolo_eval restricted to class 0 ('person', the only class yolo.py:112 keeps) + yolo.py:108-127.
    -> (boxes [n][4] int x, y, w, h; scores [n] float32), in the order detect_image returns them."""
    input_hw = (outputs[0].shape[1] * 32, outputs[0].shape[2] * 32)
    bs, ss = [], []
    for l, out in enumerate(outputs):
        b, s = boxes_and_scores(out, anchors[ANCHOR_MASK[l]], num_classes, input_hw, image_hw)
        bs.append(b)
        ss.append(s[:, 0])
    boxes, scores = np.concatenate(bs), np.concatenate(ss)
    m = scores >= F32(score_thr)
    cb, cs = boxes[m], scores[m]
    idx = tf_nms(cb, cs, max_boxes, iou_thr)
    cb, cs = cb[idx], cs[idx]
    ret_b, ret_s = [], []
    for i in reversed(range(len(cs))):
        box = cb[i]
        x, y = int(box[1]), int(box[0])
        w, h = int(box[3] - box[1]), int(box[2] - box[0])
        if x < 0:
            w, x = w + x, 0
        if y < 0:
            h, y = h + y, 0
        ret_b.append([x, y, w, h])
        ret_s.append(cs[i])
    return np.array(ret_b, np.int64).reshape(-1, 4), np.array(ret_s, F32)


def detect(model: YOLOv4Ref, frame_bgr):
    """parser.py:55-56 + yolo.detect_image on one BGR frame."""
    rgb = np.ascontiguousarray(frame_bgr[..., ::-1])
    outs = model.forward(network_input(rgb))
    return person_detections(outs, frame_bgr.shape[:2], num_classes=model.nc)
