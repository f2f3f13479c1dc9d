"""ReID appearance model of mmtrack's DeepSORT configuration as a layer program.

Spec: 3rdparty/mmtracking/mot/deepsort/deepsort_faster-rcnn_fpn_4e_mot17-private-half.py:17-42 (selected by
pose_pipeline/wrappers/mmtrack.py:16-19 for method "deepsort"): `BaseReID` = mmcls ResNet-50 (out_indices (3,), style
'pytorch') -> GlobalAveragePooling(kernel (8, 4), stride 1) -> LinearReIDHead(num_fcs 1: Linear 2048 -> 1024 + BN1d + ReLU,
fc_out: Linear 1024 -> 128); `simple_test` returns the 128-d fc_out features (the classifier / bn after it are training
only).  Checkpoint: mmtracking/checkpoints/tracktor_reid_r50_iter25245-a452f51f.pth with the names `backbone.*`,
`head.fcs.0.fc.*`, `head.fcs.0.bn.*`, `head.fc_out.*`.

Input: mmtrack's SortTracker.crop_imgs (:43-49 reid img_scale (256, 128), img_norm_cfg None) cuts every kept detection out
of the DETECTOR's normalised input tensor and resizes it to 256 x 128 with F.interpolate(bilinear, align_corners=False):
box * scale_factor in float32, clamp to the resized image, int truncation, empty sides widened by one pixel.  `crop_rects`
does the box arithmetic on the host, pp_crop_resize_bilinear the interpolation on the device, straight from the detector
program's resident input buffer.
mmtrack / mmcls are not vendored: restated from their published 0.x sources, PARITY UNPINNED (oracle: oracle/reid_mm.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ..program import Net, Program, ProgramBuilder, fold_bn
from .faster_rcnn import resnet50_body, resnet50_param_shapes

CROP_HW = (256, 128)


def reid_param_shapes() -> dict:
    sh = {}
    resnet50_param_shapes(sh, "")
    sh["head.fcs.0.fc.weight"], sh["head.fcs.0.fc.bias"] = (1024, 2048), (1024,)
    for s in ("weight", "bias", "running_mean", "running_var"):
        sh["head.fcs.0.bn." + s] = (1024,)
    sh["head.fc_out.weight"], sh["head.fc_out.bias"] = (128, 1024), (128,)
    return sh


def build_reid_program(sd: dict) -> Program:
    pb = ProgramBuilder()
    x = pb.buf(CROP_HW[0], CROP_HW[1], 4, name="input")
    c5 = resnet50_body(pb, sd, x, "")[3]                      # [8][4][2048]
    g = pb.avgpool(c5, 8, 4, 1, name="neck.gap")              # nn.AvgPool2d((8, 4), 1) -> [1][1][2048]
    w, b = fold_bn(sd["head.fcs.0.fc.weight"][:, :, None, None], sd["head.fcs.0.fc.bias"], sd["head.fcs.0.bn.weight"],
                   sd["head.fcs.0.bn.bias"], sd["head.fcs.0.bn.running_mean"], sd["head.fcs.0.bn.running_var"])
    f = pb.conv(g, w, b, relu=L.PP_RELU_LAST, name="head.fcs.0")
    pb.conv(f, sd["head.fc_out.weight"][:, :, None, None], sd["head.fc_out.bias"], out=pb.buf(1, 1, 128, name="features"),
            name="head.fc_out")
    return pb.build()


def crop_rects(boxes_xyxy, scale_factor, img_hw):
    """SortTracker.crop_imgs' box arithmetic: boxes [n][4] float32 in SOURCE pixels (the detector's rescaled output),
    scale_factor float32[4], img_hw = (h, w) of the resized (un-padded) detector input -> int32 [n][4] (x1, y1, x2, y2)"""
    b = (np.asarray(boxes_xyxy, np.float32).reshape(-1, 4) * np.asarray(scale_factor, np.float32)[None, :]).astype(np.float32)
    h, w = img_hw
    b[:, 0::2] = np.clip(b[:, 0::2], np.float32(0), np.float32(w))
    b[:, 1::2] = np.clip(b[:, 1::2], np.float32(0), np.float32(h))
    r = b.astype(np.int32)                                    # map(int, bbox): truncation
    r[:, 2] = np.where(r[:, 2] == r[:, 0], r[:, 0] + 1, r[:, 2])
    r[:, 3] = np.where(r[:, 3] == r[:, 1], r[:, 1] + 1, r[:, 3])
    # a box entirely beyond the right / bottom edge clamps to (w, w + 1): mmtrack would slice an EMPTY crop there and fail in
    # F.interpolate; here (and in the oracle) such a box takes the last pixel column / row instead
    r[:, 0], r[:, 2] = np.minimum(r[:, 0], w - 1), np.minimum(r[:, 2], w)
    r[:, 1], r[:, 3] = np.minimum(r[:, 1], h - 1), np.minimum(r[:, 3], h)
    return r


class ReidEncoder:
    """128-d appearance embeddings of detections, computed from the detector's resident input tensor"""

    def __init__(self, ctx, sd: dict, detector, max_crops: int = 128, blob_fn=None, numerics=None):
        self.ctx, self.det = ctx, detector
        self.prog = build_reid_program(sd)
        self.max_crops = int(max_crops)
        self.net = Net(ctx, self.prog, max_batch=self.max_crops, blob_dev=(blob_fn or (lambda n, p: None))("reid", self.prog), numerics=numerics)
        self.in_ptr, _, _ = self.net.buffer("input")
        h, w = detector.src
        self.sf = np.array([detector.nw / w, detector.nh / h, detector.nw / w, detector.nh / h], np.float32)

    @property
    def flops_per_crop(self):
        return self.prog.flops

    def encode(self, per_frame_boxes):
        """per_frame_boxes: for each frame of the detector's LAST run, [n][4+] float32 boxes (source pixels) -> list of
        [n][128] float32.  Must be called before the detector runs again (the crops come from its input buffer)."""
        src_ptr, _, (hp, wp, _c) = self.det.net_a.buffer("input")
        rects, owner = [], []
        for f, boxes in enumerate(per_frame_boxes):
            r = crop_rects(np.asarray(boxes, np.float32).reshape(-1, 5 if np.asarray(boxes).shape[-1] == 5 else 4)[:, :4], self.sf,
                           (self.det.nh, self.det.nw))
            for q in r:
                rects.append((f, *q))
                owner.append(f)
        out = [np.zeros((0, 128), np.float32) for _ in per_frame_boxes]
        if not rects:
            return out
        rects = np.ascontiguousarray(rects, np.int32)
        feats = np.zeros((len(rects), 128), np.float32)
        for i0 in range(0, len(rects), self.max_crops):
            part = rects[i0:i0 + self.max_crops]
            L.check(self.ctx.lib.pp_crop_resize_bilinear(self.ctx.handle, C.c_void_p(src_ptr), len(per_frame_boxes), hp, wp, L.ptr(part),
                                                         len(part), CROP_HW[0], CROP_HW[1], C.c_void_p(self.in_ptr)), "pp_crop_resize_bilinear")
            self.net.run(len(part))
            feats[i0:i0 + len(part)] = self.net.read("features", len(part)).reshape(len(part), 128)
        owner = np.asarray(owner)
        return [feats[owner == f] for f in range(len(per_frame_boxes))]
