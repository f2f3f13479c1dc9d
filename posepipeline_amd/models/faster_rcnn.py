"""Faster-RCNN R50-FPN (1 class) as two pp_net layer programs + the pp_detector handle.

Architecture spec: /root/reference/3rdparty/mmtracking/_base_/models/faster_rcnn_r50_fpn.py:1-112 with the
overrides of mot/deepsort/sort_faster-rcnn_fpn_4e_mot17-private-half.py:5-15 (num_classes=1,
clip_border=False).  Parameter names follow the mmtrack checkpoint (`detector.backbone.*`,
`detector.neck.*`, `detector.rpn_head.*`, `detector.roi_head.bbox_head.*`).

Program A (per frame):  input [Hp][Wp][4] -> ResNet-50 (style 'pytorch': stride on the 3x3) -> FPN
  (lateral 1x1 + nearest-upsampled coarser level fused into the lateral's epilogue, 3x3 output convs,
  P6 = stride-2 subsampling of P5) -> shared RPN head applied per level.
Program B (per RoI):    RoIAlign features [7][7][256] -> shared_fcs.0 as a 7x7 conv (mmdet flattens (c,h,w),
  i.e. the fc weight [1024][12544] is the conv weight [1024][256][7][7]) -> shared_fcs.1 -> fc_cls / fc_reg.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ..program import ProgramBuilder, Program, fold_bn

DET_MEAN = (123.675, 116.28, 103.53)     # _base_/datasets/mot_challenge.py:3-4
DET_STD = (58.395, 57.12, 57.375)
STRIDES = (4, 8, 16, 32, 64)
R50_LAYERS = ((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2))   # (blocks, planes, stride)


def normalize_lut(mean=DET_MEAN, std=DET_STD) -> np.ndarray:
    """mmcv.imnormalize on a float32 image: (v - mean) * (1 / std), scalars cast to float32."""
    v = np.arange(256, dtype=np.float32)
    m = np.asarray(mean, np.float64).astype(np.float32)[:, None]
    si = (1.0 / np.asarray(std, np.float64)).astype(np.float32)[:, None]
    return np.ascontiguousarray(((v[None, :] - m).astype(np.float32) * si).astype(np.float32))


def base_anchors(scales=(8,), ratios=(0.5, 1.0, 2.0), strides=STRIDES) -> np.ndarray:
    """mmdet AnchorGenerator.gen_base_anchors (center_offset 0, scale_major), float32 [levels][3][4]."""
    r = np.array(ratios, np.float32)
    s = np.array(scales, np.float32)
    hr = np.sqrt(r).astype(np.float32)
    wr = (np.float32(1) / hr).astype(np.float32)
    out = []
    for st in strides:
        ws = ((np.float32(st) * wr[:, None]).astype(np.float32) * s[None, :]).astype(np.float32).reshape(-1)
        hs = ((np.float32(st) * hr[:, None]).astype(np.float32) * s[None, :]).astype(np.float32).reshape(-1)
        out.append(np.stack([np.float32(-0.5) * ws, np.float32(-0.5) * hs, np.float32(0.5) * ws, np.float32(0.5) * hs], -1))
    return np.ascontiguousarray(np.stack(out).astype(np.float32))


def faster_rcnn_param_shapes(prefix="detector.") -> dict:
    sh = {}

    def cb(conv, bn, cout, cin, k):
        sh[prefix + conv + ".weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            sh[prefix + bn + "." + s] = (cout,)

    def cbias(name, cout, cin, k):
        sh[prefix + name + ".weight"] = (cout, cin, k, k)
        sh[prefix + name + ".bias"] = (cout,)

    cb("backbone.conv1", "backbone.bn1", 64, 3, 7)
    inpl = 64
    for li, (blocks, planes, _) in enumerate(R50_LAYERS):
        for b in range(blocks):
            q = f"backbone.layer{li + 1}.{b}."
            cb(q + "conv1", q + "bn1", planes, inpl, 1)
            cb(q + "conv2", q + "bn2", planes, planes, 3)
            cb(q + "conv3", q + "bn3", planes * 4, planes, 1)
            if b == 0:
                cb(q + "downsample.0", q + "downsample.1", planes * 4, inpl, 1)
            inpl = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        cbias(f"neck.lateral_convs.{i}.conv", 256, c, 1)
        cbias(f"neck.fpn_convs.{i}.conv", 256, 256, 3)
    cbias("rpn_head.rpn_conv", 256, 256, 3)
    cbias("rpn_head.rpn_cls", 3, 256, 1)
    cbias("rpn_head.rpn_reg", 12, 256, 1)
    h = prefix + "roi_head.bbox_head."
    sh[h + "shared_fcs.0.weight"], sh[h + "shared_fcs.0.bias"] = (1024, 12544), (1024,)
    sh[h + "shared_fcs.1.weight"], sh[h + "shared_fcs.1.bias"] = (1024, 1024), (1024,)
    sh[h + "fc_cls.weight"], sh[h + "fc_cls.bias"] = (2, 1024), (2,)
    sh[h + "fc_reg.weight"], sh[h + "fc_reg.bias"] = (4, 1024), (4,)
    return sh


def resnet50_param_shapes(sh: dict, prefix: str):
    """conv + BatchNorm parameters of a torchvision-style ResNet-50 (`backbone.*` names of mmdet / mmcls checkpoints)"""
    def cb(conv, bn, cout, cin, k):
        sh[prefix + conv + ".weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            sh[prefix + bn + "." + s] = (cout,)

    cb("backbone.conv1", "backbone.bn1", 64, 3, 7)
    inpl = 64
    for li, (blocks, planes, _) in enumerate(R50_LAYERS):
        for b in range(blocks):
            q = f"backbone.layer{li + 1}.{b}."
            cb(q + "conv1", q + "bn1", planes, inpl, 1)
            cb(q + "conv2", q + "bn2", planes, planes, 3)
            cb(q + "conv3", q + "bn3", planes * 4, planes, 1)
            if b == 0:
                cb(q + "downsample.0", q + "downsample.1", planes * 4, inpl, 1)
            inpl = planes * 4


def resnet50_body(pb, sd: dict, x, prefix: str):
    """ResNet-50, style 'pytorch' (stride on the 3x3), BatchNorm folded: -> [C2, C3, C4, C5]"""
    R = L.PP_RELU_LAST
    p = prefix

    def cb(x, conv, bn, **kw):
        w, b = fold_bn(sd[p + conv + ".weight"], None, sd[p + bn + ".weight"], sd[p + bn + ".bias"],
                       sd[p + bn + ".running_mean"], sd[p + bn + ".running_var"])
        return pb.conv(x, w, b, name=conv, **kw)

    x = cb(x, "backbone.conv1", "backbone.bn1", stride=2, pad=3, relu=R)
    x = pb.maxpool(x, 3, 2, 1)
    feats = []
    for li, (blocks, planes, stride) in enumerate(R50_LAYERS):
        for b in range(blocks):
            q = f"backbone.layer{li + 1}.{b}."
            s = stride if b == 0 else 1
            idn = cb(x, q + "downsample.0", q + "downsample.1", stride=s) if b == 0 else x
            y = cb(x, q + "conv1", q + "bn1", relu=R)
            y = cb(y, q + "conv2", q + "bn2", stride=s, pad=1, relu=R)
            x = cb(y, q + "conv3", q + "bn3", relu=R, res1=idn)
        feats.append(x)
    return feats


def build_image_program(sd: dict, hp: int, wp: int, prefix="detector.") -> Program:
    pb = ProgramBuilder()
    R = L.PP_RELU_LAST
    p = prefix

    def cbias(x, name, **kw):
        return pb.conv(x, sd[p + name + ".weight"], sd[p + name + ".bias"], name=name, **kw)

    x = pb.buf(hp, wp, 4, name="input")
    feats = resnet50_body(pb, sd, x, prefix)
    # FPN: lat[3] = conv(C5); lat[i-1] = conv(C_{i-1}) + nearest_up(lat[i])  (fused as a shifted residual read)
    lat = [None] * 4
    lat[3] = cbias(feats[3], "neck.lateral_convs.3.conv")
    for i in range(2, -1, -1):
        lat[i] = cbias(feats[i], f"neck.lateral_convs.{i}.conv", res1=lat[i + 1], res1_shift=1)
    outs = []
    for i in range(4):
        h, w, _ = pb.dims(lat[i])
        outs.append(cbias(lat[i], f"neck.fpn_convs.{i}.conv", pad=1, out=pb.buf(h, w, 256, name=f"p{i + 2}")))
    outs.append(pb.maxpool(outs[3], 1, 2, 0, name="p6"))
    # RPN head: rpn_cls (3 anchors) and rpn_reg (3 x 4 deltas) as ONE 1x1 convolution of 15 output channels into a 16-channel map
    # per level (objectness 0 - 2, deltas 3 - 14): the 256-channel rpn_conv output is read once instead of twice (at 160x272 that read
    # is 2.85 GB per 64 frames; the two heads were 1.5 ms of the step, HBM-bound).  Every output channel is its own fmaf chain over k,
    # so the values are the two separate convolutions' bit for bit (tests/test_gpu_detector.py compares them with the oracle's maps).
    w_rpn = np.concatenate([sd[p + "rpn_head.rpn_cls.weight"], sd[p + "rpn_head.rpn_reg.weight"]], axis=0)
    b_rpn = np.concatenate([sd[p + "rpn_head.rpn_cls.bias"], sd[p + "rpn_head.rpn_reg.bias"]], axis=0)
    assert w_rpn.shape[:2] == (15, 256)
    # a zero 16th output channel: every channel of the 16-channel map is WRITTEN (reproducible buffer dumps, whole float4 stores);
    # rpn_select never reads it
    w_rpn = np.concatenate([w_rpn, np.zeros((1,) + w_rpn.shape[1:], w_rpn.dtype)], axis=0)
    b_rpn = np.concatenate([b_rpn, np.zeros(1, b_rpn.dtype)], axis=0)
    for l, f in enumerate(outs):
        h, w, _ = pb.dims(f)
        t = cbias(f, "rpn_head.rpn_conv", pad=1, relu=R)
        pb.conv(t, w_rpn, b_rpn, name="rpn_head.rpn_cls+rpn_reg", out=pb.buf(h, w, 16, name=f"rpn{l}"))
    return pb.build()


def build_roi_program(sd: dict, prefix="detector.") -> Program:
    pb = ProgramBuilder()
    h = prefix + "roi_head.bbox_head."
    x = pb.buf(7, 7, 256, name="roi_in")
    x = pb.conv(x, sd[h + "shared_fcs.0.weight"].reshape(1024, 256, 7, 7), sd[h + "shared_fcs.0.bias"], relu=L.PP_RELU_LAST,
                name="shared_fcs.0")
    x = pb.conv(x, sd[h + "shared_fcs.1.weight"][:, :, None, None], sd[h + "shared_fcs.1.bias"], relu=L.PP_RELU_LAST,
                name="shared_fcs.1")
    pb.conv(x, sd[h + "fc_cls.weight"][:, :, None, None], sd[h + "fc_cls.bias"], out=pb.buf(1, 1, 2, name="cls"), name="fc_cls")
    pb.conv(x, sd[h + "fc_reg.weight"][:, :, None, None], sd[h + "fc_reg.bias"], out=pb.buf(1, 1, 4, name="reg"), name="fc_reg")
    return pb.build()


def detector_input_size(src_h, src_w):
    v = [C.c_int32() for _ in range(4)]
    L.check(L.load_library().pp_detector_input_size(src_h, src_w, *[C.byref(x) for x in v]), "pp_detector_input_size")
    nh, nw, hp, wp = (x.value for x in v)
    return nh, nw, hp, wp


class Detector:
    """pp_detector handle: frames [F][H][W][3] u8 BGR -> per-frame [n][5] (x1, y1, x2, y2, score)."""

    MAX_ROIS = 1000
    MAX_DET = 100

    def __init__(self, ctx, sd, src_h, src_w, max_frames=4, prefix="detector.", blob_fn=None, numerics=None):
        """blob_fn(name, program) -> (device pointer, n_floats) or None: where the program's weight blob is already resident
        on the device ("det_a" = image program, "det_b" = RoI head; RCCL broadcast, posepipeline_amd/parallel.py)"""
        from ..program import Net
        blob_fn = blob_fn or (lambda name, prog: None)
        self.ctx = ctx
        self.src = (src_h, src_w)
        self.nh, self.nw, self.hp, self.wp = detector_input_size(src_h, src_w)
        self.prog_a = build_image_program(sd, self.hp, self.wp, prefix)
        self.prog_b = build_roi_program(sd, prefix)
        self.net_a = Net(ctx, self.prog_a, max_batch=max_frames, blob_dev=blob_fn("det_a", self.prog_a), numerics=numerics)
        self.net_b = Net(ctx, self.prog_b, max_batch=max_frames * self.MAX_ROIS, blob_dev=blob_fn("det_b", self.prog_b), numerics=numerics)
        na = self.prog_a.named
        # (cls_l == reg_l: the fused RPN head's 16-channel map, see pp_detector_create)
        bufs_a = np.array([na["input"]] + [na[f"rpn{l}"] for l in range(5)] + [na[f"rpn{l}"] for l in range(5)] +
                          [na[f"p{i}"] for i in range(2, 6)], np.int32)
        nb = self.prog_b.named
        bufs_b = np.array([nb["roi_in"], nb["cls"], nb["reg"]], np.int32)
        h = C.c_void_p()
        lut, anchors = normalize_lut(), base_anchors()      # keep the arrays alive across the call
        L.check(ctx.lib.pp_detector_create(self.net_a.handle, self.net_b.handle, L.ptr(bufs_a), L.ptr(bufs_b), src_h, src_w,
                                           L.ptr(lut), L.ptr(anchors), C.byref(h)), "pp_detector_create")
        self.handle = h
        self.max_frames = max_frames

    @property
    def flops_per_frame(self):
        return self.prog_a.flops + self.MAX_ROIS * self.prog_b.flops

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):
                self.ctx.lib.pp_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    MARGIN_NAMES = ("rpn_cut", "rpn_nms", "rpn_top", "roi_level", "score_thr", "det_nms", "det_top", "det_order")

    def enable_margins(self, enable=True, score_weight=8.0):
        """per-frame decision margins with every run (include/posepipe_hip.h: pp_detector_enable_margins); read them with
        `margins()` after `run`.  score_weight: IoU error bound / (2 x score error bound) of the numerics to be certified."""
        L.check(self.ctx.lib.pp_detector_enable_margins(self.handle, int(bool(enable)), C.c_float(score_weight)), "pp_detector_enable_margins")
        self._margins_on = bool(enable)

    def margins(self, n_frames):
        """[n_frames][8] float32 (MARGIN_NAMES) of the most recent run"""
        m = np.zeros((n_frames, len(self.MARGIN_NAMES)), np.float32)
        L.check(self.ctx.lib.pp_detector_margins(self.handle, n_frames, L.ptr(m)), "pp_detector_margins")
        return m

    def run(self, frames, want_proposals=False, frames_dev=None):
        """frames: numpy [F][H][W][3] u8 BGR (or frames_dev=(ptr, F) for device-resident frames)."""
        if frames_dev is not None:
            ptr, f = frames_dev
            fptr, mem = C.c_void_p(ptr), L.PP_MEM_DEVICE
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            f = frames.shape[0]
            assert frames.shape[1:] == (self.src[0], self.src[1], 3)
            fptr, mem = L.ptr(frames), L.PP_MEM_HOST
        dets = np.zeros((f, self.MAX_DET, 5), np.float32)
        n = np.zeros((f,), np.int32)
        props = np.zeros((f, self.MAX_ROIS, 4), np.float32) if want_proposals else None
        nprops = np.zeros((f,), np.int32) if want_proposals else None
        L.check(self.ctx.lib.pp_detector_run(self.handle, fptr, f, mem, L.ptr(dets), L.ptr(n), L.ptr(props), L.ptr(nprops)),
                "pp_detector_run")
        out = [dets[i, : n[i]].copy() for i in range(f)]
        if want_proposals:
            return out, [props[i, : nprops[i]].copy() for i in range(f)]
        return out

    def enqueue(self, frames, want_proposals=False, frames_dev=None):
        """the pass of `run`, queued on the context's stream without waiting (pp_detector_enqueue); `collect()` returns what `run`
        would have.  One pass in flight; host `frames` must stay alive until the collect."""
        if frames_dev is not None:
            ptr, f = frames_dev
            fptr, mem, keep = C.c_void_p(ptr), L.PP_MEM_DEVICE, None
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            f = frames.shape[0]
            assert frames.shape[1:] == (self.src[0], self.src[1], 3)
            fptr, mem, keep = L.ptr(frames), L.PP_MEM_HOST, frames
        L.check(self.ctx.lib.pp_detector_enqueue(self.handle, fptr, f, mem, int(bool(want_proposals))), "pp_detector_enqueue")
        self._inflight = (f, want_proposals, keep)

    def collect(self):
        if getattr(self, "_inflight", None) is None:
            raise L.PosePipeHipError("pp_detector_collect: no pass in flight")
        f, want_proposals, _keep = self._inflight
        self._inflight = None
        dets = np.zeros((f, self.MAX_DET, 5), np.float32)
        n = np.zeros((f,), np.int32)
        props = np.zeros((f, self.MAX_ROIS, 4), np.float32) if want_proposals else None
        nprops = np.zeros((f,), np.int32) if want_proposals else None
        L.check(self.ctx.lib.pp_detector_collect(self.handle, L.ptr(dets), L.ptr(n), L.ptr(props), L.ptr(nprops)), "pp_detector_collect")
        out = [dets[i, : n[i]].copy() for i in range(f)]
        if want_proposals:
            return out, [props[i, : nprops[i]].copy() for i in range(f)]
        return out

    def timing(self):
        ms = np.zeros(6, np.float32)
        L.check(self.ctx.lib.pp_detector_timing(self.handle, L.ptr(ms)), "pp_detector_timing")
        return dict(zip(("preprocess", "image_program", "rpn_proposals", "roi_align", "roi_head", "final"), map(float, ms)))
