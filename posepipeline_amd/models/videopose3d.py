"""VideoPose3D temporal lifting network as a pp_net layer program (whole-clip dilated form).

The reference builds `TemporalModelOptimized1f(17, 2, 17, filter_widths=[3,3,3,3,3], causal=False,
dropout=0.25, channels=1024)` (pose_pipeline/wrappers/videopose3d.py:10-16,46-50) and feeds it one
243-frame window per output frame.  The strided model is arithmetically the dilated `TemporalModel`
with dilations 1, 3, 9, 27, 81 evaluated at every frame, so this program runs the dilated form over
chunks of `chunk` output frames (+121-frame halos) with the same state_dict (`expand_conv`,
`expand_bn`, `layers_conv.{0..7}`, `layers_bn.{0..7}`, `shrink`).

Activations are (B, H=1, W=time, C) NHWC; the 34 input features are padded to 36 channels.
Block i:  x = res[centre crop] + relu(bn(conv1x1(relu(bn(conv_k3_dilated(x))))))  -- the residual is the
input of the block cropped by the dilation on both sides (`res1_off_w`), ReLU before the add
(PP_RELU_FIRST).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .. import _lib as L
from ..program import ProgramBuilder, Program, fold_bn


@dataclass(frozen=True)
class VideoPose3DSpec:
    num_joints_in: int = 17
    in_features: int = 2
    num_joints_out: int = 17
    filter_widths: tuple = (3, 3, 3, 3, 3)
    channels: int = 1024
    chunk: int = 512             # output frames per program invocation

    @property
    def receptive_field(self):
        rf = 1
        for f in self.filter_widths:
            rf *= f
        return rf

    @property
    def pad(self):
        return (self.receptive_field - 1) // 2


def videopose3d_param_shapes(spec: VideoPose3DSpec) -> dict:
    sh = {}
    cin = spec.num_joints_in * spec.in_features
    ch = spec.channels
    sh["expand_conv.weight"] = (ch, cin, spec.filter_widths[0])
    for s in ("weight", "bias", "running_mean", "running_var"):
        sh["expand_bn." + s] = (ch,)
    for i in range(len(spec.filter_widths) - 1):
        sh[f"layers_conv.{2 * i}.weight"] = (ch, ch, spec.filter_widths[i + 1])
        sh[f"layers_conv.{2 * i + 1}.weight"] = (ch, ch, 1)
        for j in (2 * i, 2 * i + 1):
            for s in ("weight", "bias", "running_mean", "running_var"):
                sh[f"layers_bn.{j}.{s}"] = (ch,)
    sh["shrink.weight"] = (spec.num_joints_out * 3, ch, 1)
    sh["shrink.bias"] = (spec.num_joints_out * 3,)
    return sh


def build_videopose3d_program(spec: VideoPose3DSpec, sd: dict) -> Program:
    for k, shp in videopose3d_param_shapes(spec).items():
        if k not in sd:
            raise KeyError(f"missing parameter {k}")
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {sd[k].shape} != {shp}")
    pb = ProgramBuilder()
    cin = spec.num_joints_in * spec.in_features
    cin_p = (cin + 3) // 4 * 4
    t_in = spec.chunk + 2 * spec.pad

    def cb(x, conv, bn, dil, **kw):
        w, b = fold_bn(sd[conv + ".weight"], None, sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                       sd[bn + ".running_var"])
        return pb.conv(x, w, b, dil=(1, dil), name=conv, **kw)

    x = pb.buf(1, t_in, cin_p, name="input")
    x = cb(x, "expand_conv", "expand_bn", 1, relu=L.PP_RELU_LAST)
    dil = spec.filter_widths[0]
    for i in range(len(spec.filter_widths) - 1):
        y = cb(x, f"layers_conv.{2 * i}", f"layers_bn.{2 * i}", dil, relu=L.PP_RELU_LAST)
        x = cb(y, f"layers_conv.{2 * i + 1}", f"layers_bn.{2 * i + 1}", 1, relu=L.PP_RELU_FIRST, res1=x, res1_off_w=dil)
        dil *= spec.filter_widths[i + 1]
    out = pb.buf(1, spec.chunk, spec.num_joints_out * 3, name="output")
    pb.conv(x, sd["shrink.weight"], sd["shrink.bias"], out=out, name="shrink")
    return pb.build()
