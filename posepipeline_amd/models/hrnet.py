"""HRNet-W{32,48} top-down pose network as a pp_net layer program.

Architecture spec: /root/reference/3rdparty/mmpose/config/top_down/darkpose/coco/
hrnet_w48_coco_384x288_dark.py:44-79 (backbone `extra` stage table, TopDownSimpleHead with
num_deconv_layers=0, final_conv_kernel=1).  W32 is the same family with widths (32,64,128,256)
(BASELINE.json configs 1-2).  Parameter names follow mmpose 0.x (`backbone.*`,
`keypoint_head.final_layer.*`) so that a real checkpoint's state_dict can be packed unchanged.

Fusion done here (the reference runs every conv / BN / ReLU / add / upsample as its own op):
  * BN folded into each conv; ReLU and the residual add in the conv epilogue;
  * HRModule fuse layers  y_i = relu(sum_j f_ij(x_j))  are accumulated in mmpose's j order: the strided terms (j < i) and
    the identity term by the epilogues of the strided convs (residual adds), the coarse terms (j > i) -- 1x1 conv + BN at
    the COARSE resolution, nearest upsample -- by one PP_OP_UPSAMPLE_ADD pass over the fine map that adds all of them in
    order and applies the ReLU: one read and one write of the fine map per output instead of one per term.  Same additions
    in the same order as mmpose's `y += ...` loop, so the result is bit-identical to the unfused evaluation.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .. import _lib as L
from ..program import ProgramBuilder, Program, fold_bn

import os

# How the coarse-to-fine terms of a fuse layer are accumulated (A/B switch, see _HR.module): "onepass" (default) = the 1x1 convs
# stay at their own resolution and ONE PP_OP_UPSAMPLE_ADD pass adds all of them to the partial sum; "conv" = round 1's form,
# each 1x1 conv's epilogue scatters over its 2^u x 2^u patch (one read + write of the fine map per term).  Same bits either way.
FUSE_MODE = os.environ.get("POSEPIPE_HRNET_FUSE", "onepass")

COCO_FLIP_PAIRS = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16)]
# Halpe-136: derived from the `swap=` fields of /root/reference/3rdparty/mmpose/config/_base_/halpe.py
HALPE_FLIP_PAIRS = COCO_FLIP_PAIRS + [
    (20, 21), (22, 23), (24, 25), (26, 42), (27, 41), (28, 40), (29, 39), (30, 38), (31, 37), (32, 36), (33, 35),
    (43, 52), (44, 51), (45, 50), (46, 49), (47, 48), (57, 61), (58, 60), (62, 71), (63, 70), (64, 69), (65, 68),
    (66, 73), (67, 72), (74, 80), (75, 79), (76, 78), (81, 85), (82, 84), (86, 90), (87, 89), (91, 93)
] + [(94 + i, 115 + i) for i in range(21)]
# COCO-WholeBody-133: the dataset file the vendored config points to (_base_/datasets/coco_wholebody.py) is NOT in the
# reference tree; pairs restated from the published dataset definition (unpinned).
WHOLEBODY_FLIP_PAIRS = COCO_FLIP_PAIRS + [(17, 20), (18, 21), (19, 22)] + [(23 + i, 39 - i) for i in range(8)] + [
    (40, 49), (41, 48), (42, 47), (43, 46), (44, 45), (54, 58), (55, 57), (59, 68), (60, 67), (61, 66), (62, 65),
    (63, 70), (64, 69), (71, 77), (72, 76), (73, 75), (78, 82), (79, 81), (83, 87), (84, 86), (88, 90)
] + [(91 + i, 112 + i) for i in range(21)]


@dataclass(frozen=True)
class HRNetSpec:
    width: int = 48
    num_joints: int = 17
    in_h: int = 384
    in_w: int = 288
    # (num_modules, num_branches) of stages 2..4, 4 BasicBlocks per branch; stage1 = 4 Bottlenecks(64)
    stages: tuple = ((1, 2), (4, 3), (3, 4))
    blocks_per_branch: int = 4

    @property
    def channels(self):
        return tuple(self.width * (2 ** i) for i in range(4))

    @property
    def heatmap_hw(self):
        return self.in_h // 4, self.in_w // 4


def hrnet_w48_384x288(num_joints=17):
    return HRNetSpec(48, num_joints, 384, 288)


def hrnet_w32_256x192(num_joints=17):
    return HRNetSpec(32, num_joints, 256, 192)


def flip_perm(num_joints=17, pairs=COCO_FLIP_PAIRS):
    perm = np.arange(num_joints, dtype=np.int32)
    for a, b in pairs:
        perm[a], perm[b] = b, a
    return perm


# ---- parameter inventory (names + shapes), used for synthetic init and checkpoint validation ----
def _conv_bn(shapes, conv, bn, cout, cin, k):
    shapes[conv + ".weight"] = (cout, cin, k, k)
    for s in ("weight", "bias", "running_mean", "running_var"):
        shapes[bn + "." + s] = (cout,)


def hrnet_param_shapes(spec: HRNetSpec) -> dict:
    sh: dict = {}
    B = "backbone."
    _conv_bn(sh, B + "conv1", B + "bn1", 64, 3, 3)
    _conv_bn(sh, B + "conv2", B + "bn2", 64, 64, 3)
    for i in range(4):
        p = f"{B}layer1.{i}."
        cin = 64 if i == 0 else 256
        _conv_bn(sh, p + "conv1", p + "bn1", 64, cin, 1)
        _conv_bn(sh, p + "conv2", p + "bn2", 64, 64, 3)
        _conv_bn(sh, p + "conv3", p + "bn3", 256, 64, 1)
        if i == 0:
            _conv_bn(sh, p + "downsample.0", p + "downsample.1", 256, 64, 1)
    ch = spec.channels
    pre = [256]
    for si, (n_mod, n_br) in enumerate(spec.stages):
        stage = si + 2
        cur = list(ch[:n_br])
        t = f"{B}transition{si + 1}."
        for i in range(n_br):
            if i < len(pre):
                if pre[i] != cur[i]:
                    _conv_bn(sh, f"{t}{i}.0", f"{t}{i}.1", cur[i], pre[i], 3)
            else:
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = cur[i] if j == i - len(pre) else cin
                    _conv_bn(sh, f"{t}{i}.{j}.0", f"{t}{i}.{j}.1", cout, cin, 3)
        for m in range(n_mod):
            mp = f"{B}stage{stage}.{m}."
            for b in range(n_br):
                for k in range(spec.blocks_per_branch):
                    p = f"{mp}branches.{b}.{k}."
                    _conv_bn(sh, p + "conv1", p + "bn1", cur[b], cur[b], 3)
                    _conv_bn(sh, p + "conv2", p + "bn2", cur[b], cur[b], 3)
            last = (si == len(spec.stages) - 1) and (m == n_mod - 1)
            n_out = 1 if last else n_br
            for i in range(n_out):
                for j in range(n_br):
                    f = f"{mp}fuse_layers.{i}.{j}."
                    if j > i:
                        _conv_bn(sh, f + "0", f + "1", cur[i], cur[j], 1)
                    elif j < i:
                        for k in range(i - j):
                            cout = cur[i] if k == i - j - 1 else cur[j]
                            _conv_bn(sh, f"{f}{k}.0", f"{f}{k}.1", cout, cur[j], 3)
        pre = cur
    sh["keypoint_head.final_layer.weight"] = (spec.num_joints, ch[0], 1, 1)
    sh["keypoint_head.final_layer.bias"] = (spec.num_joints,)
    return sh


# ---- program builder -----------------------------------------------------------------------------
class _HR:
    def __init__(self, spec, sd):
        self.spec, self.sd, self.pb = spec, sd, ProgramBuilder()

    def cb(self, x, conv, bn, *, stride=1, pad=1, relu=L.PP_RELU_NONE, **kw):
        sd = self.sd
        w, b = fold_bn(sd[conv + ".weight"], None, sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                       sd[bn + ".running_var"])
        return self.pb.conv(x, w, b, stride=stride, pad=pad, relu=relu, name=conv, **kw)

    def bottleneck(self, x, p, has_ds):
        R = L.PP_RELU_LAST
        idn = self.cb(x, p + "downsample.0", p + "downsample.1", pad=0) if has_ds else x
        y = self.cb(x, p + "conv1", p + "bn1", pad=0, relu=R)
        y = self.cb(y, p + "conv2", p + "bn2", pad=1, relu=R)
        return self.cb(y, p + "conv3", p + "bn3", pad=0, relu=R, res1=idn)

    def basic(self, x, p):
        R = L.PP_RELU_LAST
        y = self.cb(x, p + "conv1", p + "bn1", relu=R)
        return self.cb(y, p + "conv2", p + "bn2", relu=R, res1=x)

    def module(self, xs, mp, n_out):
        spec, pb = self.spec, self.pb
        n_br = len(xs)
        xs = list(xs)
        for b in range(n_br):
            for k in range(spec.blocks_per_branch):
                xs[b] = self.basic(xs[b], f"{mp}branches.{b}.{k}.")
        outs = []
        for i in range(n_out):
            # terms T_j in mmpose's order j = 0..n_br-1:  y = ((T_0 + T_1) + T_2) + ...; relu(y).
            # A conv term takes the partial sum as res1 and, when the identity term x_i directly
            # follows it (j + 1 == i), x_i as res2:  (partial + conv) + x_i.
            acc = -1
            j = 0
            ups = []                                    # "onepass": the coarse terms (buffer, up_log2), j > i, added at the end
            onepass = FUSE_MODE == "onepass"
            while j < n_br:
                if j == i:
                    assert acc == -1 and i == 0
                    acc = xs[i]                         # y = 0 + x_0
                    j += 1
                    continue
                absorb = (j + 1 == i)
                is_last = (j == n_br - 1) or (absorb and i == n_br - 1)
                relu = L.PP_RELU_LAST if is_last else L.PP_RELU_NONE
                f = f"{mp}fuse_layers.{i}.{j}."
                if j > i:
                    if onepass:
                        ups.append((self.cb(xs[j], f + "0", f + "1", pad=0), j - i))      # 1x1 conv + BN at the coarse resolution
                    else:
                        # the conv epilogue scatters (partial + value) over the 2^u x 2^u patch
                        h, w, c = pb.dims(xs[i])
                        acc = self.cb(xs[j], f + "0", f + "1", pad=0, relu=relu, res1=acc, up_log2=j - i,
                                      out=pb.buf(h, w, c))
                else:
                    y = xs[j]
                    for k in range(i - j - 1):
                        y = self.cb(y, f"{f}{k}.0", f"{f}{k}.1", stride=2, relu=L.PP_RELU_LAST)
                    k = i - j - 1
                    acc = self.cb(y, f"{f}{k}.0", f"{f}{k}.1", stride=2, relu=relu, res1=acc,
                                  res2=xs[i] if absorb else -1)
                j += 2 if absorb else 1
            if ups:
                # y = relu(((acc + up(t_a)) + up(t_b)) + up(t_c)): every coarse term of this output in one pass
                acc = pb.upsample_add(ups[0][0], up_log2=ups[0][1], res1=acc, relu=L.PP_RELU_LAST, more=ups[1:],
                                      name=f"{mp}fuse_layers.{i}.up")
            outs.append(acc)
        return outs

    def build(self) -> Program:
        spec, pb = self.spec, self.pb
        R = L.PP_RELU_LAST
        B = "backbone."
        x = pb.buf(spec.in_h, spec.in_w, 4, name="input")     # RGB + one zero channel (Cin % 4 == 0)
        x = self.cb(x, B + "conv1", B + "bn1", stride=2, relu=R)
        x = self.cb(x, B + "conv2", B + "bn2", stride=2, relu=R)
        for i in range(4):
            x = self.bottleneck(x, f"{B}layer1.{i}.", i == 0)
        ch = spec.channels
        ys = [x]
        pre = [256]
        for si, (n_mod, n_br) in enumerate(spec.stages):
            cur = list(ch[:n_br])
            t = f"{B}transition{si + 1}."
            xs = []
            for i in range(n_br):
                if i < len(pre):
                    if pre[i] != cur[i]:
                        xs.append(self.cb(ys[i], f"{t}{i}.0", f"{t}{i}.1", relu=R))
                    else:
                        xs.append(ys[i])
                else:
                    y = ys[-1]
                    for j in range(i + 1 - len(pre)):
                        y = self.cb(y, f"{t}{i}.{j}.0", f"{t}{i}.{j}.1", stride=2, relu=R)
                    xs.append(y)
            for m in range(n_mod):
                last = (si == len(spec.stages) - 1) and (m == n_mod - 1)
                xs = self.module(xs, f"{B}stage{si + 2}.{m}.", 1 if last else n_br)
            ys = xs
            pre = cur
        hh, hw = spec.heatmap_hw
        out = pb.buf(hh, hw, spec.num_joints, name="output")
        sd = self.sd
        pb.conv(ys[0], sd["keypoint_head.final_layer.weight"], sd["keypoint_head.final_layer.bias"], pad=0, out=out,
                out_nchw=True, name="keypoint_head.final_layer")
        return pb.build()


def build_hrnet_program(spec: HRNetSpec, state_dict: dict) -> Program:
    """state_dict: name -> numpy array in torch layouts (see hrnet_param_shapes)."""
    shapes = hrnet_param_shapes(spec)
    for k, shp in shapes.items():
        if k not in state_dict:
            raise KeyError(f"missing parameter {k}")
        if tuple(state_dict[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {state_dict[k].shape} != {shp}")
    return _HR(spec, state_dict).build()
