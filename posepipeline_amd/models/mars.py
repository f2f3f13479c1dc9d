"""mars-small128 appearance encoder of the DeepSortYOLOv4 tracking method as a layer program.

Network: pose_pipeline/wrappers/deep_sort_yolov4/tools/freeze_model.py:119-229 (conv1_1, conv1_2, max_pool 3x3/2
VALID, six pre-activation residual blocks, fc1, 'ball' batch norm, L2 normalisation); crops:
tools/generate_detections.py:25-63 (extract_image_patch) and :92-105 (create_box_encoder).

Mapping to the program: conv + slim.batch_norm (epsilon 1e-3, no gamma) + ELU fold into one conv with an ELU epilogue;
a standalone batch norm (+ ELU) -- the pre-activation of a residual link, 'ball' -- is a 1x1 convolution with a
diagonal weight, which computes fl(fl(x*s) + b) exactly; TensorFlow SAME padding puts the odd row / column last
(`pad_end`); fc1 over the flattened (h, w, c) map is a 16x8 'valid' convolution; the L2 normalisation of the 128
features runs on the host.  mars-small128.pb is a TensorFlow GraphDef (no parser here): seeded weights only.
Parameter names: <scope>.weight [cout][cin][kh][kw] (fc1.weight [128][16*8*128]), <scope>.bias where slim creates one
(the second conv of a block), <scope>.bn.{beta,mean,var}, ball.{beta,mean,var}.
"""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..program import Net, Program, ProgramBuilder, fold_bn

BN_EPS = 1e-3
PATCH_HW = (128, 64)
BLOCKS = (("conv2_1", 32, False, True), ("conv2_3", 32, False, False), ("conv3_1", 64, True, False),
          ("conv3_3", 64, False, False), ("conv4_1", 128, True, False), ("conv4_3", 128, False, False))


def mars_param_shapes() -> dict:
    sh = {}

    def bn(name, c):
        for s in ("beta", "mean", "var"):
            sh[f"{name}.{s}"] = (c,)

    sh["conv1_1.weight"] = (32, 3, 3, 3)
    bn("conv1_1.bn", 32)
    sh["conv1_2.weight"] = (32, 32, 3, 3)
    bn("conv1_2.bn", 32)
    cin = 32
    for scope, c, inc, first in BLOCKS:
        if not first:
            bn(scope + ".bn", cin)
        sh[scope + ".1.weight"] = (c, cin, 3, 3)
        bn(scope + ".1.bn", c)
        sh[scope + ".2.weight"] = (c, c, 3, 3)
        sh[scope + ".2.bias"] = (c,)
        if inc:
            sh[scope + ".projection.weight"] = (c, cin, 1, 1)
        cin = c
    sh["fc1.weight"] = (128, 16 * 8 * 128)
    bn("fc1.bn", 128)
    bn("ball", 128)
    return sh


def tf_same(n: int, k: int, stride: int):
    """TensorFlow SAME -> (pad_begin, extra_at_end in {0, 1}, out)"""
    out = -(-n // stride)
    total = max((out - 1) * stride + k - n, 0)
    begin = total // 2
    assert total - begin - begin in (0, 1)
    return begin, total - 2 * begin, out


def build_mars_program(sd: dict) -> Program:
    pb = ProgramBuilder()
    ones = lambda c: np.ones(c, np.float32)

    def conv(x, name, stride=1, bn=True, act=L.PP_ACT_ELU, **kw):
        w = sd[name + ".weight"]
        b = sd.get(name + ".bias")
        if bn:
            c = w.shape[0]
            w, b = fold_bn(w, b, ones(c), sd[name + ".bn.beta"], sd[name + ".bn.mean"], sd[name + ".bn.var"], BN_EPS)
        h, wd, _ = pb.dims(x)
        k = w.shape[2]
        ph, eh, _ = tf_same(h, k, stride)
        pw, ew, _ = tf_same(wd, k, stride)
        return pb.conv(x, w, b, stride=stride, pad=(ph, pw), pad_end=(eh, ew), relu=act, name=name, **kw)

    def bn_act(x, prefix, act):
        """standalone slim.batch_norm (+ ELU) as a diagonal 1x1 convolution"""
        c = pb.dims(x)[2]
        var = sd[prefix + ".var"].astype(np.float64)
        s = 1.0 / np.sqrt(var + BN_EPS)
        b = sd[prefix + ".beta"].astype(np.float64) - sd[prefix + ".mean"].astype(np.float64) * s
        w = np.zeros((c, c, 1, 1), np.float32)
        w[np.arange(c), np.arange(c), 0, 0] = s.astype(np.float32)
        return pb.conv(x, w, b.astype(np.float32), relu=act, name=prefix)

    x = pb.buf(PATCH_HW[0], PATCH_HW[1], 4, name="input")
    x = conv(x, "conv1_1")
    x = conv(x, "conv1_2")
    x = pb.maxpool(x, 3, 2, 0, name="pool1")                        # VALID
    for scope, c, inc, first in BLOCKS:
        net = x if first else bn_act(x, scope + ".bn", L.PP_ACT_ELU)
        y = conv(net, scope + ".1", stride=2 if inc else 1)
        short = conv(x, scope + ".projection", stride=2, bn=False, act=L.PP_RELU_NONE) if inc else x
        x = conv(y, scope + ".2", bn=False, act=L.PP_RELU_NONE, res1=short)
    h, w, c = pb.dims(x)
    wfc = sd["fc1.weight"].reshape(-1, h, w, c).transpose(0, 3, 1, 2)
    wf, bf = fold_bn(np.ascontiguousarray(wfc), sd.get("fc1.bias"), ones(wfc.shape[0]), sd["fc1.bn.beta"], sd["fc1.bn.mean"],
                     sd["fc1.bn.var"], BN_EPS)
    f = pb.conv(x, wf, bf, relu=L.PP_ACT_ELU, name="fc1")
    out = pb.buf(1, 1, 128, name="features")
    c128 = 128
    var = sd["ball.var"].astype(np.float64)
    s = 1.0 / np.sqrt(var + BN_EPS)
    b = sd["ball.beta"].astype(np.float64) - sd["ball.mean"].astype(np.float64) * s
    wd = np.zeros((c128, c128, 1, 1), np.float32)
    wd[np.arange(c128), np.arange(c128), 0, 0] = s.astype(np.float32)
    pb.conv(f, wd, b.astype(np.float32), relu=L.PP_RELU_NONE, out=out, name="ball")
    return pb.build()


def patch_rect(bbox_tlwh, image_hw, patch_hw=PATCH_HW):
    """extract_image_patch's box arithmetic (generate_detections.py:44-60) -> (sx, sy, ex, ey) or None.
    The dtype of `bbox_tlwh` is kept (yolo.detect_image returns Python ints -> integer arithmetic with truncation)."""
    bbox = np.array(bbox_tlwh)
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


class MarsEncoder:
    """create_box_encoder: (frames, boxes) -> unit-norm 128-d features (float64 rows, generate_detections.py:80)."""

    def __init__(self, ctx: L.Context, sd: dict, src_h: int, src_w: int, max_patches: int = 64, numerics=None):
        self.ctx, self.src = ctx, (src_h, src_w)
        self.prog = build_mars_program(sd)
        self.net = Net(ctx, self.prog, max_batch=max_patches, numerics=numerics)
        self.max_patches = max_patches

    def encode(self, frames, boxes_per_frame, frames_dev=None):
        """boxes_per_frame: per frame [m][4] (x, y, w, h) as yolo.detect_image returns them.
        -> per frame [m][128] float64.  A box whose patch is empty is encoded from an all-zero patch."""
        rects, owner = [], []
        for f, boxes in enumerate(boxes_per_frame):
            for b in boxes:
                r = patch_rect(b, self.src)
                if r is None:
                    # the reference prints a warning and encodes an UNSEEDED random patch (generate_detections.py:98-101);
                    # an all-zero patch stands in for it here (empty rect -> zeros in pp_reid_patches)
                    r = (0, 0, 0, 0)
                rects.append((f,) + r)
                owner.append(f)
        n = len(rects)
        feats = np.zeros((n, 128), np.float32)
        if frames_dev is not None:
            ptr, nf = frames_dev
            src, mem = L.ptr(int(ptr)), L.PP_MEM_DEVICE
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            nf, src, mem = frames.shape[0], L.ptr(frames), L.PP_MEM_HOST
        din, _, _ = self.net.buffer("input")
        dout, _, _ = self.net.buffer("features")
        for s0 in range(0, n, self.max_patches):
            part = np.ascontiguousarray(np.array(rects[s0:s0 + self.max_patches], np.int32))
            L.check(self.ctx.lib.pp_reid_patches(self.ctx.handle, src, nf, self.src[0], self.src[1], mem, L.ptr(part), len(part),
                                                 PATCH_HW[0], PATCH_HW[1], L.ptr(int(din))), "pp_reid_patches")
            self.net.run(len(part))
            chunk = np.empty((len(part), 128), np.float32)
            self.ctx.d2h(chunk, int(dout))
            feats[s0:s0 + len(part)] = chunk
        # features / sqrt(1e-8 + sum(features^2)): float32, sequential accumulation
        sq = feats * feats
        acc = np.cumsum(sq, axis=1, dtype=np.float32)[:, -1] if n else np.zeros(0, np.float32)
        norm = np.sqrt(np.float32(1e-8) + acc).astype(np.float32)
        feats = (feats / norm[:, None]).astype(np.float64)
        out, k = [], 0
        for boxes in boxes_per_frame:
            out.append(feats[k:k + len(boxes)])
            k += len(boxes)
        return out

    def close(self):
        self.net.close()
