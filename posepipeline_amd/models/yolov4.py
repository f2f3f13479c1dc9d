"""YOLOv4 person detector of the DeepSortYOLOv4 tracking method as a layer program.

Architecture: pose_pipeline/wrappers/deep_sort_yolov4/yolo4/model.py:55-190 (DarknetConv2D_BN_Mish / _Leaky,
resblock_body, darknet_body, yolo4_body); inference wrapper: wrappers/deep_sort_yolov4/yolo.py:85-129 (letterbox to
416x416, / 255, yolo_eval with score 0.5 / IoU 0.5, 'person' only, int-truncated boxes).

Mapping to the program: Conv+BN folds into (weight, bias) with Keras' epsilon 1e-3; Mish / LeakyReLU(0.1) are conv
epilogues; every Concatenate is a wider buffer whose producers write channel slices (`out_c_off`); UpSampling2D(2) is the
2x2 scatter epilogue of the 1x1 conv before it; the SPP max pools read and write slices of one 2048-channel buffer;
ZeroPadding2D(((1,0),(1,0))) + 'valid' stride-2 convs equal a symmetric pad of 1 for the even sizes of a 416 input.
Parameters are named l0, l1, ... in the order yolo4_body creates its convolutions:
l{i}.weight [cout][cin][k][k], l{i}.bn.{gamma,beta,mean,var} or l{i}.bias (the three 1x1 output convs).
yolo4.h5 is a Keras/HDF5 checkpoint; h5py is not available in the build image, so only seeded weights are exercised.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ..program import Program, ProgramBuilder, fold_bn, Net

# model_data/yolo_anchors.txt (not in the repository): the anchors keras-yolo4 ships
ANCHORS = np.array([[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]], np.float32)
ANCHOR_MASK = [[6, 7, 8], [3, 4, 5], [0, 1, 2]]
BN_EPS = 1e-3
PIL_PRECISION_BITS = 32 - 8 - 2


# ---- layer list in creation order ---------------------------------------------------------------------------
def _layers(num_classes=80):
    """(cin, cout, k, has_bn) of every convolution in the order yolo4_body creates them."""
    out = []

    def conv(cin, cout, k, bn=True):
        out.append((cin, cout, k, bn))
        return cout

    def resblock(cin, filters, blocks, all_narrow=True):
        half = filters // 2 if all_narrow else filters
        conv(cin, filters, 3)
        conv(filters, half, 1)           # shortconv
        conv(filters, half, 1)           # mainconv
        for _ in range(blocks):
            conv(half, filters // 2, 1)
            conv(filters // 2, half, 3)
        conv(half, half, 1)              # postconv
        return conv(2 * half, filters, 1)

    c = conv(3, 32, 3)
    c = resblock(c, 64, 1, False)
    c = resblock(c, 128, 2)
    c76 = c = resblock(c, 256, 8)
    c38 = c = resblock(c, 512, 8)
    c = resblock(c, 1024, 4)
    nout = 3 * (num_classes + 5)
    for co, k in ((512, 1), (1024, 3), (512, 1)):
        c = conv(c, co, k)
    c = 4 * 512
    for co, k in ((512, 1), (1024, 3), (512, 1)):
        c = conv(c, co, k)
    conv(512, 256, 1)                    # y19_upsample
    conv(c38, 256, 1)
    c = 512
    for co, k in ((256, 1), (512, 3), (256, 1), (512, 3), (256, 1)):
        c = conv(c, co, k)
    conv(256, 128, 1)                    # y38_upsample
    conv(c76, 128, 1)
    c = 256
    for co, k in ((128, 1), (256, 3), (128, 1), (256, 3), (128, 1)):
        c = conv(c, co, k)
    conv(128, 256, 3)
    conv(256, nout, 1, bn=False)         # y76_output
    conv(128, 256, 3)                    # y76_downsample (stride 2)
    c = 512
    for co, k in ((256, 1), (512, 3), (256, 1), (512, 3), (256, 1)):
        c = conv(c, co, k)
    conv(256, 512, 3)
    conv(512, nout, 1, bn=False)         # y38_output
    conv(256, 512, 3)                    # y38_downsample (stride 2)
    c = 1024
    for co, k in ((512, 1), (1024, 3), (512, 1), (1024, 3), (512, 1)):
        c = conv(c, co, k)
    conv(512, 1024, 3)
    conv(1024, nout, 1, bn=False)        # y19_output
    return out


def yolov4_param_shapes(num_classes=80) -> dict:
    shapes = {}
    for i, (cin, cout, k, bn) in enumerate(_layers(num_classes)):
        shapes[f"l{i}.weight"] = (cout, cin, k, k)
        if bn:
            for s in ("gamma", "beta", "mean", "var"):
                shapes[f"l{i}.bn.{s}"] = (cout,)
        else:
            shapes[f"l{i}.bias"] = (cout,)
    return shapes


def synth_params(shapes: dict, seed: int = 0, head_bias: float | None = None) -> dict:
    """Seeded parameters for names ending in .weight / .bias / .gamma / .beta / .mean / .var (He-normal weights)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shp in shapes.items():
        if name.endswith(".var"):
            a = rng.uniform(0.5, 1.5, shp)
        elif name.endswith(".mean") or name.endswith(".beta"):
            a = rng.normal(0, 0.1, shp)
        elif name.endswith(".gamma"):
            a = rng.uniform(0.5, 1.0, shp)
        elif name.endswith(".bias"):
            a = rng.normal(0, 0.1, shp) if head_bias is None else rng.normal(head_bias, 1.0, shp)
        else:
            a = rng.normal(0, np.sqrt(2.0 / int(np.prod(shp[1:]))), shp)
        sd[name] = a.astype(np.float32)
    return sd


def seed_person_head(sd: dict, num_classes: int = 80, anchor: int = 0, logit: float = 3.0) -> dict:
    """Make seeded weights produce person candidates (a random head yields none or thousands): on the coarsest grid,
    objectness and class-0 logits of one anchor get a +logit bias, every other objectness bias -6.  In place."""
    heads = [k for k in sd if k.endswith(".bias")]          # creation order: y76_output, y38_output, y19_output
    assert len(heads) == 3
    per = 5 + num_classes
    for name in heads:
        b = sd[name]
        b[4::per] = -6.0
    b = sd[heads[-1]]
    b[anchor * per + 2: anchor * per + 4] = 0.0             # w, h = the anchor's
    b[anchor * per + 4] = logit
    b[anchor * per + 5] = logit
    return sd


def yolov4_macs(size=416, num_classes=80) -> float:
    """analytic MACs of one forward pass (computed while building the program: Program.flops / 2)"""
    return build_yolov4_program(synth_params(yolov4_param_shapes(num_classes), 0), size, num_classes).flops / 2


# ---- program --------------------------------------------------------------------------------------------------
def build_yolov4_program(sd: dict, size: int = 416, num_classes: int = 80) -> Program:
    assert size % 32 == 0
    pb = ProgramBuilder()
    idx = [0]

    def conv(x, stride=1, act=L.PP_ACT_MISH, **kw):
        i = idx[0]
        idx[0] += 1
        w = sd[f"l{i}.weight"]
        if f"l{i}.bias" in sd:
            wf, bf = w, sd[f"l{i}.bias"]
        else:
            wf, bf = fold_bn(w, None, sd[f"l{i}.bn.gamma"], sd[f"l{i}.bn.beta"], sd[f"l{i}.bn.mean"], sd[f"l{i}.bn.var"], BN_EPS)
        k = w.shape[2]
        return pb.conv(x, wf, bf, stride=stride, pad=k // 2, relu=act, name=f"l{i}", **kw)

    def resblock(x, filters, blocks, all_narrow=True):
        half = filters // 2 if all_narrow else filters
        pre = conv(x, 2)
        h, w, _ = pb.dims(pre)
        cat = pb.buf(h, w, 2 * half)                     # Concatenate([postconv, shortconv])
        conv(pre, out=cat, out_c_off=half)
        main = conv(pre)
        for _ in range(blocks):
            y = conv(main)
            main = conv(y, res1=main)                    # Add()([mainconv, y]): y = mish(conv) comes first, then + main
        conv(main, out=cat, out_c_off=0)
        return conv(cat)

    LK = L.PP_ACT_LEAKY
    x = pb.buf(size, size, 4, name="input")
    x = conv(x)
    x = resblock(x, 64, 1, False)
    x = resblock(x, 128, 2)
    f76 = x = resblock(x, 256, 8)
    f38 = x = resblock(x, 512, 8)
    x = resblock(x, 1024, 4)
    g = size // 32
    # SPP: Concatenate([maxpool13, maxpool9, maxpool5, y19]) lives in one buffer
    y = conv(conv(x, act=LK), act=LK)
    spp = pb.buf(g, g, 2048)
    conv(y, act=LK, out=spp, out_c_off=1536)
    for j, k in enumerate((13, 9, 5)):
        pb.maxpool(spp, k, 1, k // 2, name=f"spp{k}", out=spp, out_c_off=512 * j, in_c_off=1536, c=512)
    y19 = conv(conv(conv(spp, act=LK), act=LK), act=LK)
    cat38 = pb.buf(2 * g, 2 * g, 512)                    # Concatenate([conv(f38), upsample(conv(y19))])
    conv(y19, act=LK, up_log2=1, out=cat38, out_c_off=256)
    conv(f38, act=LK, out=cat38, out_c_off=0)
    y38 = cat38
    for _ in range(5):
        y38 = conv(y38, act=LK)
    cat76 = pb.buf(4 * g, 4 * g, 256)
    conv(y38, act=LK, up_log2=1, out=cat76, out_c_off=128)
    conv(f76, act=LK, out=cat76, out_c_off=0)
    y76 = cat76
    for _ in range(5):
        y76 = conv(y76, act=LK)
    o76 = pb.buf(4 * g, 4 * g, 3 * (num_classes + 5), name="y76")
    conv(conv(y76, act=LK), act=L.PP_RELU_NONE, out=o76)
    cat38b = pb.buf(2 * g, 2 * g, 512)                   # Concatenate([y76_downsample, y38])
    conv(y76, 2, act=LK, out=cat38b, out_c_off=0)
    pb.maxpool(y38, 1, 1, 0, name="route38", out=cat38b, out_c_off=256)      # slice copy
    y38 = cat38b
    for _ in range(5):
        y38 = conv(y38, act=LK)
    o38 = pb.buf(2 * g, 2 * g, 3 * (num_classes + 5), name="y38")
    conv(conv(y38, act=LK), act=L.PP_RELU_NONE, out=o38)
    cat19b = pb.buf(g, g, 1024)                          # Concatenate([y38_downsample, y19])
    conv(y38, 2, act=LK, out=cat19b, out_c_off=0)
    pb.maxpool(y19, 1, 1, 0, name="route19", out=cat19b, out_c_off=512)
    y19 = cat19b
    for _ in range(5):
        y19 = conv(y19, act=LK)
    o19 = pb.buf(g, g, 3 * (num_classes + 5), name="y19")
    conv(conv(y19, act=LK), act=L.PP_RELU_NONE, out=o19)
    assert idx[0] == len(_layers(num_classes)), (idx[0], len(_layers(num_classes)))
    return pb.build()


# ---- PIL bicubic coefficient tables (Pillow precompute_coeffs + normalize_coeffs_8bpc) -------------------------
def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_table(in_size: int, out_size: int):
    """-> (int32 [out_size][2 + ksize] rows (first source index, tap count, coefficients << 22), ksize)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    tab = np.zeros((out_size, 2 + ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = [_bicubic((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        tot = 0.0
        for w in ws:
            tot += w
        if tot != 0.0:
            ws = [w / tot for w in ws]
        tab[xx, 0], tab[xx, 1] = xmin, xmax
        for x, w in enumerate(ws):
            tab[xx, 2 + x] = int(-0.5 + w * (1 << PIL_PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PIL_PRECISION_BITS))
    return tab, ksize


def letterbox_geometry(src_h: int, src_w: int, size: int = 416):
    """letterbox_image's (nw, nh) (yolo4/utils.py:23-27)"""
    scale = min(size / src_w, size / src_h)
    return int(src_w * scale), int(src_h * scale)


class YoloV4Detector:
    """yolo.YOLO.detect_image on chunks of frames: letterbox -> program -> decode -> NMS -> person boxes."""

    def __init__(self, ctx: L.Context, sd: dict, src_h: int, src_w: int, max_frames: int = 8, size: int = 416,
                 num_classes: int = 80, score: float = 0.5, iou: float = 0.5, max_boxes: int = 200, anchors=ANCHORS, numerics=None):
        self.ctx, self.size, self.nc = ctx, size, num_classes
        self.src = (src_h, src_w)
        self.score, self.iou, self.max_boxes = score, iou, max_boxes
        self.anchors = np.asarray(anchors, np.float32)
        self.prog = build_yolov4_program(sd, size, num_classes)
        self.net = Net(ctx, self.prog, max_batch=max_frames, numerics=numerics)
        self.max_frames = max_frames
        self.nw, self.nh = letterbox_geometry(src_h, src_w, size)
        self.xtab, self.kx = pil_bicubic_table(src_w, self.nw)
        self.ytab, self.ky = pil_bicubic_table(src_h, self.nh)

    @property
    def flops_per_frame(self):
        return self.prog.flops

    def preprocess(self, frames, frames_dev=None):
        if frames_dev is not None:
            ptr, n = frames_dev
            src, mem = L.ptr(int(ptr)), L.PP_MEM_DEVICE
        else:
            frames = np.ascontiguousarray(frames, np.uint8)
            n, src, mem = frames.shape[0], L.ptr(frames), L.PP_MEM_HOST
            assert frames.shape[1:] == (self.src[0], self.src[1], 3), frames.shape
        assert 0 < n <= self.max_frames
        dptr, _, _ = self.net.buffer("input")
        L.check(self.ctx.lib.pp_letterbox_bicubic(self.ctx.handle, src, n, self.src[0], self.src[1], mem, L.ptr(self.xtab),
                                                  self.nw, self.kx, L.ptr(self.ytab), self.nh, self.ky, self.size, self.size,
                                                  L.ptr(int(dptr))), "pp_letterbox_bicubic")
        return n

    def decode(self, n):
        """-> per frame (boxes [m][4] int64 x, y, w, h; scores [m] float32) in yolo.detect_image's order"""
        bs, ss = [], []
        for l, name in enumerate(("y19", "y38", "y76")):
            dptr, _, _ = self.net.buffer(name)
            g = self.size // 32 << l
            boxes = np.empty((n, g * g * 3, 4), np.float32)
            scores = np.empty((n, g * g * 3), np.float32)
            anc = np.ascontiguousarray(self.anchors[ANCHOR_MASK[l]])
            L.check(self.ctx.lib.pp_yolo_decode(self.ctx.handle, L.ptr(int(dptr)), n, g, g, self.nc, 0, L.ptr(anc), self.size,
                                                self.size, self.src[0], self.src[1], L.ptr(boxes), L.ptr(scores), L.PP_MEM_HOST),
                    "pp_yolo_decode")
            bs.append(boxes)
            ss.append(scores)
        boxes, scores = np.concatenate(bs, 1), np.concatenate(ss, 1)
        out = []
        for f in range(n):
            m = scores[f] >= np.float32(self.score)
            cb, cs = np.ascontiguousarray(boxes[f][m]), np.ascontiguousarray(scores[f][m])
            keep = np.zeros(max(len(cs), 1), np.int32)
            nk = C.c_int32(0)
            L.check(self.ctx.lib.pp_nms(self.ctx.handle, L.ptr(cb), L.ptr(cs), len(cs), float(self.iou), 2, L.ptr(keep),
                                        C.byref(nk), L.PP_MEM_HOST), "pp_nms")
            keep = keep[: min(nk.value, self.max_boxes)]
            rb, rs = [], []
            for i in reversed(keep):                    # yolo.py:108: reversed(list(enumerate(out_classes)))
                box = cb[i]
                x, y = int(box[1]), int(box[0])
                w, h = int(box[3] - box[1]), int(box[2] - box[0])
                if x < 0:
                    w, x = w + x, 0
                if y < 0:
                    h, y = h + y, 0
                rb.append([x, y, w, h])
                rs.append(cs[i])
            out.append((np.array(rb, np.int64).reshape(-1, 4), np.array(rs, np.float32)))
        return out

    def run(self, frames, frames_dev=None):
        n = self.preprocess(frames, frames_dev)
        self.ctx.timer_start()
        self.net.run(n)
        self.last_net_ms = self.ctx.timer_stop()       # HIP events around the program (bench stage times)
        return self.decode(n)

    def close(self):
        self.net.close()
