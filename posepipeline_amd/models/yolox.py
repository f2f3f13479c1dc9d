"""YOLOX-X person detector of mmtracking's ByteTrack configuration as a layer program.

Selected by pose_pipeline/wrappers/mmtrack.py:8-29 with method "bytetrack":
3rdparty/mmtracking/mot/bytetrack/bytetrack_yolox_x_crowdhuman_mot17-private-half.py:9-20 over
3rdparty/mmtracking/_base_/models/yolox_x_8x8.py:4-27 -- CSPDarknet(deepen 1.33, widen 1.25), YOLOXPAFPN(in [320, 640, 1280],
out 320, 4 CSP blocks), YOLOXHead(1 class), test scale (800, 1440), score_thr 0.01, NMS IoU 0.7; test pipeline :62-78
(Resize keep_ratio, Normalize mean 0 / std 1 without channel swap, Pad to 32 with 114).  mmdet is not vendored: the
module internals follow mmdet 2.x (PARITY UNPINNED, see oracle/yolox.py).  Parameter names are mmdet's state_dict keys.

Mapping to the program: ConvModule = conv + BN(eps 1e-3) folded + Swish epilogue; Focus = a 2x2 stride-2 convolution
with 0/1 weights (exact) that also undoes the BGR order of the decoded frame (the wrapper hands mmtrack an RGB array and
the pipeline does not swap again); every torch.cat is a wider buffer written in channel slices; SPP pools slices of one
buffer; nearest 2x upsampling is the scatter epilogue of a (repeated) 1x1 reduce convolution.  The head decode
(priors, exp, sigmoid, score filter) runs on the host in float32 with double-evaluated transcendentals; NMS is pp_nms.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ..program import Net, Program, ProgramBuilder, fold_bn

BN_EPS = 1e-3
STRIDES = (8, 16, 32)
WIDEN, DEEPEN = 1.25, 1.33
PAD_VALUE = 114.0            # Pad(size_divisor=32, pad_val=dict(img=(114.0, 114.0, 114.0))), bytetrack_*.py test pipeline
SIZE_DIVISOR = 32
ARCH = ((64, 128, 3, True, False), (128, 256, 9, True, False), (256, 512, 9, True, False), (512, 1024, 3, False, True))


def _cm(sh, name, cin, cout, k):
    sh[name + ".conv.weight"] = (cout, cin, k, k)
    for s in ("weight", "bias", "running_mean", "running_var"):
        sh[f"{name}.bn.{s}"] = (cout,)


def _csp(sh, name, cin, cout, blocks):
    mid = int(cout * 0.5)
    _cm(sh, name + ".main_conv", cin, mid, 1)
    _cm(sh, name + ".short_conv", cin, mid, 1)
    _cm(sh, name + ".final_conv", 2 * mid, cout, 1)
    for b in range(blocks):
        _cm(sh, f"{name}.blocks.{b}.conv1", mid, mid, 1)
        _cm(sh, f"{name}.blocks.{b}.conv2", mid, mid, 3)


def yolox_param_shapes(num_classes=1, prefix="detector.") -> dict:
    sh = {}
    P = prefix
    _cm(sh, P + "backbone.stem.conv", 12, int(64 * WIDEN), 3)
    for i, (cin, cout, nb, _ident, spp) in enumerate(ARCH):
        cin, cout, nb = int(cin * WIDEN), int(cout * WIDEN), max(round(nb * DEEPEN), 1)
        s = f"{P}backbone.stage{i + 1}"
        _cm(sh, s + ".0", cin, cout, 3)
        j = 1
        if spp:
            _cm(sh, s + ".1.conv1", cout, cout // 2, 1)
            _cm(sh, s + ".1.conv2", cout // 2 * 4, cout, 1)
            j = 2
        _csp(sh, f"{s}.{j}", cout, cout, nb)
    inc = [320, 640, 1280]
    for k, idx in enumerate((2, 1)):
        _cm(sh, f"{P}neck.reduce_layers.{k}", inc[idx], inc[idx - 1], 1)
        _csp(sh, f"{P}neck.top_down_blocks.{k}", inc[idx - 1] * 2, inc[idx - 1], 4)
    for idx in (0, 1):
        _cm(sh, f"{P}neck.downsamples.{idx}", inc[idx], inc[idx], 3)
        _csp(sh, f"{P}neck.bottom_up_blocks.{idx}", inc[idx] * 2, inc[idx + 1], 4)
    for i in range(3):
        _cm(sh, f"{P}neck.out_convs.{i}", inc[i], 320, 1)
    for l in range(3):
        for j in range(2):
            _cm(sh, f"{P}bbox_head.multi_level_cls_convs.{l}.{j}", 320, 320, 3)
            _cm(sh, f"{P}bbox_head.multi_level_reg_convs.{l}.{j}", 320, 320, 3)
        for name, c in (("cls", num_classes), ("reg", 4), ("obj", 1)):
            sh[f"{P}bbox_head.multi_level_conv_{name}.{l}.weight"] = (c, 320, 1, 1)
            sh[f"{P}bbox_head.multi_level_conv_{name}.{l}.bias"] = (c,)
    return sh


def seed_synthetic_head(sd: dict, obj_bias: float = -7.0, prefix="detector.") -> dict:
    """Make seeded weights behave like a detector with few candidates: the stem absorbs the 0..255 input scale and the
    objectness bias is strongly negative (a random head lets half of the 22 050 priors through).  In place."""
    sd[prefix + "backbone.stem.conv.conv.weight"] *= np.float32(1.0 / 255.0)
    for l in range(3):
        sd[f"{prefix}bbox_head.multi_level_conv_obj.{l}.bias"][:] = obj_bias
        sd[f"{prefix}bbox_head.multi_level_conv_cls.{l}.bias"][:] = 2.0
    return sd


def build_yolox_program(sd: dict, hp: int, wp: int, prefix="detector.") -> Program:
    assert hp % 32 == 0 and wp % 32 == 0
    pb = ProgramBuilder()
    P = prefix
    SW = L.PP_ACT_SWISH

    def cm(x, name, stride=1, **kw):
        p = P + name
        w, b = fold_bn(sd[p + ".conv.weight"], None, sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"],
                       sd[p + ".bn.running_var"], BN_EPS)
        return pb.conv(x, w, b, stride=stride, pad=w.shape[2] // 2, relu=SW, name=name, **kw)

    def csp(x, name, blocks, ident, **kw):
        h, w, _ = pb.dims(x)
        mid = sd[P + name + ".main_conv.conv.weight"].shape[0]
        cat = pb.buf(h, w, 2 * mid)                       # torch.cat((x_main, x_short), dim=1)
        cm(x, name + ".short_conv", out=cat, out_c_off=mid)
        main = cm(x, name + ".main_conv")
        for b in range(blocks):
            y = cm(main, f"{name}.blocks.{b}.conv1")
            last = b == blocks - 1
            main = cm(y, f"{name}.blocks.{b}.conv2", res1=main if ident else -1, **(dict(out=cat, out_c_off=0) if last else {}))
        return cm(cat, name + ".final_conv", **kw)

    def slice_copy(src, dst, off):
        pb.maxpool(src, 1, 1, 0, name="route", out=dst, out_c_off=off)

    x = pb.buf(hp, wp, 4, name="input")
    # Focus (patches TL, BL, TR, BR on the channel axis) as an exact 0/1 convolution; input channel 2-c of the BGR
    # frame feeds colour c of the RGB tensor the reference network sees
    fw = np.zeros((12, 4, 2, 2), np.float32)
    for p, (kh, kw) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        for c in range(3):
            fw[3 * p + c, 2 - c, kh, kw] = 1.0
    x = pb.conv(x, fw, None, stride=2, pad=0, name="focus")
    x = cm(x, "backbone.stem.conv")
    feats = []
    for i, (_cin, _cout, nb, ident, spp) in enumerate(ARCH):
        nb = max(round(nb * DEEPEN), 1)
        s = f"backbone.stage{i + 1}"
        x = cm(x, s + ".0", stride=2)
        j = 1
        if spp:
            h, w, c = pb.dims(x)
            cat = pb.buf(h, w, c // 2 * 4)                # torch.cat([x, pool5, pool9, pool13])
            cm(x, s + ".1.conv1", out=cat, out_c_off=0)
            for n, k in enumerate((5, 9, 13)):
                pb.maxpool(cat, k, 1, k // 2, name=f"spp{k}", out=cat, out_c_off=(n + 1) * (c // 2), in_c_off=0, c=c // 2)
            x = cm(cat, s + ".1.conv2")
            j = 2
        x = csp(x, f"{s}.{j}", nb, ident)
        if i >= 1:
            feats.append(x)
    inner = [feats[-1]]
    for k, idx in enumerate((2, 1)):
        low = feats[idx - 1]
        h, w, c = pb.dims(low)
        cat = pb.buf(h, w, 2 * c)                         # torch.cat([upsample(feat_high), feat_low])
        high = cm(inner[0], f"neck.reduce_layers.{k}")
        cm(inner[0], f"neck.reduce_layers.{k}", up_log2=1, out=cat, out_c_off=0)     # the same conv, scattered 2x2
        slice_copy(low, cat, c)
        inner[0] = high
        inner.insert(0, csp(cat, f"neck.top_down_blocks.{k}", 4, False))
    outs = [inner[0]]
    for idx in (0, 1):
        hi = inner[idx + 1]
        h, w, c = pb.dims(hi)
        cl = pb.dims(outs[-1])[2]
        cat = pb.buf(h, w, cl + c)                        # torch.cat([downsample(feat_low), feat_high])
        cm(outs[-1], f"neck.downsamples.{idx}", stride=2, out=cat, out_c_off=0)
        slice_copy(hi, cat, cl)
        outs.append(csp(cat, f"neck.bottom_up_blocks.{idx}", 4, False))
    for l, o in enumerate(outs):
        f = cm(o, f"neck.out_convs.{l}")
        c = r = f
        for j in range(2):
            c = cm(c, f"bbox_head.multi_level_cls_convs.{l}.{j}")
            r = cm(r, f"bbox_head.multi_level_reg_convs.{l}.{j}")
        h, w, _ = pb.dims(f)
        for name, src in (("cls", c), ("reg", r), ("obj", r)):
            p = f"{P}bbox_head.multi_level_conv_{name}.{l}"
            out = pb.buf(h, w, sd[p + ".weight"].shape[0], name=f"{name}{l}")
            pb.conv(src, sd[p + ".weight"], sd[p + ".bias"], out=out, name=f"conv_{name}.{l}")
    return pb.build()


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


class YoloXDetector:
    """mmdet YOLOX.simple_test(rescale=True) on chunks of frames: resize / pad -> program -> decode -> NMS."""

    def __init__(self, ctx: L.Context, sd: dict, src_h: int, src_w: int, max_frames: int = 4, scale=(800, 1440),
                 score_thr: float = 0.01, iou_thr: float = 0.7, numerics=None):
        self.ctx, self.src = ctx, (src_h, src_w)
        dims = [C.c_int32() for _ in range(4)]
        L.check(ctx.lib.pp_rescale_size(src_h, src_w, max(scale), min(scale), SIZE_DIVISOR, *[C.byref(d) for d in dims]), "pp_rescale_size")
        self.nh, self.nw, self.hp, self.wp = (int(d.value) for d in dims)
        self.scale_factor = np.array([self.nw / src_w, self.nh / src_h, self.nw / src_w, self.nh / src_h], np.float32)
        self.score_thr, self.iou_thr = score_thr, iou_thr
        self.prog = build_yolox_program(sd, self.hp, self.wp)
        self.net = Net(ctx, self.prog, max_batch=max_frames, numerics=numerics)
        self.max_frames = max_frames
        self.lut = np.ascontiguousarray(np.tile(np.arange(256, dtype=np.float32), (3, 1)))     # mean 0, std 1
        self.priors = []
        for s in STRIDES:
            h, w = self.hp // s, self.wp // s
            xx = np.tile(np.arange(w, dtype=np.float32) * np.float32(s), h)
            yy = np.repeat(np.arange(h, dtype=np.float32) * np.float32(s), w)
            self.priors.append(np.stack([xx, yy, np.full_like(xx, s), np.full_like(xx, s)], -1))
        self.priors = np.concatenate(self.priors)

    @property
    def flops_per_frame(self):
        return self.prog.flops

    def run(self, frames, frames_dev=None):
        """-> per frame [n][5] float32 (x1, y1, x2, y2, score) in source pixels, descending score"""
        if frames_dev is not None:
            ptr, n = frames_dev
            src, mem = L.ptr(int(ptr)), L.PP_MEM_DEVICE
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, src, mem = frames.shape[0], L.ptr(frames), L.PP_MEM_HOST
        assert 0 < n <= self.max_frames
        din, _, _ = self.net.buffer("input")
        L.check(self.ctx.lib.pp_resize_pad_normalize(self.ctx.handle, src, n, self.src[0], self.src[1], mem, self.nh, self.nw,
                                                     self.hp, self.wp, L.ptr(self.lut), PAD_VALUE, L.ptr(int(din))),
                "pp_resize_pad_normalize")
        self.ctx.timer_start()
        self.net.run(n)
        self.last_net_ms = self.ctx.timer_stop()
        maps = {}
        for name in ("cls", "reg", "obj"):
            per_level = []
            for l, s in enumerate(STRIDES):
                dptr, _, _ = self.net.buffer(f"{name}{l}")
                c = 4 if name == "reg" else (1 if name == "obj" else self.prog.bufs[self.prog.named[f"cls{l}"]][2])
                a = np.empty((n, (self.hp // s) * (self.wp // s), c), np.float32)
                self.ctx.d2h(a, int(dptr))
                per_level.append(a)
            maps[name] = np.concatenate(per_level, 1)
        out = []
        for f in range(n):
            cls, reg, obj = _sigmoid(maps["cls"][f]), maps["reg"][f], _sigmoid(maps["obj"][f][:, 0])
            xys = reg[:, :2] * self.priors[:, 2:] + self.priors[:, :2]
            whs = np.exp(reg[:, 2:].astype(np.float64)).astype(np.float32) * self.priors[:, 2:]
            half = whs / np.float32(2)
            boxes = np.concatenate([xys - half, xys + half], 1) / self.scale_factor[None]
            max_scores = cls.max(1)
            valid = (obj * max_scores) >= np.float32(self.score_thr)
            b, s = np.ascontiguousarray(boxes[valid], np.float32), np.ascontiguousarray(max_scores[valid] * obj[valid], np.float32)
            keep = np.zeros(max(len(s), 1), np.int32)
            nk = C.c_int32(0)
            if len(s) > 8192:          # pp_nms capacity: keep the best 8192 candidates (only random weights get here)
                top = np.argsort(-s, kind="stable")[:8192]
                b, s = np.ascontiguousarray(b[top]), np.ascontiguousarray(s[top])
            L.check(self.ctx.lib.pp_nms(self.ctx.handle, L.ptr(b), L.ptr(s), len(s), float(self.iou_thr), 0, L.ptr(keep), C.byref(nk),
                                        L.PP_MEM_HOST), "pp_nms")
            keep = keep[: nk.value]
            out.append(np.concatenate([b[keep], s[keep, None]], 1).astype(np.float32))
        return out

    def close(self):
        self.net.close()
