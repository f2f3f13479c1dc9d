"""ViTPose (ViT backbone + two-deconvolution heatmap head) as a pp_net layer program.

BASELINE.json configs[4] names "ViTPose-H backbone (bf16 MFMA path)".  ViTPose is NOT in the reference tree
(out of contract); a PosePipe user would load it through the same two calls as the HRNet models
(`init_pose_model` wrappers/mmpose.py:57, `inference_top_down_pose_model` :75), so it plugs into the same top-down slot
here.  Architecture and state_dict key names follow the published ViTPose code (mmpose 0.x fork):

  backbone  ViT: PatchEmbed Conv2d(3, dim, 16, stride 16, padding 2) -> + pos_embed[:, 1:] + pos_embed[:, :1] ->
            depth x [x += proj(attn(norm1(x))); x += fc2(gelu(fc1(norm2(x))))] -> last_norm   (LayerNorm eps 1e-6)
  head      TopdownHeatmapSimpleHead: 2 x [ConvTranspose2d(4, 2, 1, bias False) + BN + ReLU] + Conv2d(256, K, 1)
  test cfg  flip_test, UDP (`use_udp=True`): unbiased affine crop + DARK-UDP decode, Gaussian modulate kernel 11

Program: patch embedding = one fp32 implicit-GEMM convolution (exact), the encoder = PP_OP_VIT_ENCODER (bf16 MFMA),
the first deconvolution = one bf16 GEMM over its 16 kernel taps + a gather (PP_OP_DECONV_BF16), the second = four fp32 2x2
convolutions (one per output parity) + depth_to_space, final 1x1 conv -> NCHW maps.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .. import _lib as L
from ..program import ProgramBuilder, Program, fold_bn


@dataclass(frozen=True)
class VitPoseSpec:
    dim: int = 1280
    depth: int = 32
    heads: int = 16
    mlp_ratio: int = 4
    num_joints: int = 17
    in_h: int = 256
    in_w: int = 192
    deconv: tuple = (256, 256)
    patch: int = 16
    patch_pad: int = 2
    # the first deconvolution (dim -> 256, 1 GMAC per pass) as one bf16 GEMM over the 16 kernel taps (PP_OP_DECONV_BF16)
    # instead of four exact fp32 convolutions: 2.2 -> 0.5 ms per 128 passes; the second one (K = 256) stays fp32
    head_bf16: bool = True

    @property
    def grid(self):
        return ((self.in_h + 2 * self.patch_pad - self.patch) // self.patch + 1,
                (self.in_w + 2 * self.patch_pad - self.patch) // self.patch + 1)

    @property
    def tokens(self):
        return self.grid[0] * self.grid[1]

    @property
    def heatmap_hw(self):
        return self.grid[0] * 4, self.grid[1] * 4


def vitpose_huge(num_joints=17):
    return VitPoseSpec(1280, 32, 16, 4, num_joints)


def vitpose_large(num_joints=17):
    return VitPoseSpec(1024, 24, 16, 4, num_joints)


def vitpose_base(num_joints=17):
    return VitPoseSpec(768, 12, 12, 4, num_joints)


def vitpose_param_shapes(spec: VitPoseSpec) -> dict:
    d, hid = spec.dim, spec.dim * spec.mlp_ratio
    s = {"backbone.patch_embed.proj.weight": (d, 3, spec.patch, spec.patch), "backbone.patch_embed.proj.bias": (d,),
         "backbone.pos_embed": (1, spec.tokens + 1, d)}
    for i in range(spec.depth):
        k = f"backbone.blocks.{i}."
        s.update({k + "norm1.weight": (d,), k + "norm1.bias": (d,), k + "attn.qkv.weight": (3 * d, d),
                  k + "attn.qkv.bias": (3 * d,), k + "attn.proj.weight": (d, d), k + "attn.proj.bias": (d,),
                  k + "norm2.weight": (d,), k + "norm2.bias": (d,), k + "mlp.fc1.weight": (hid, d),
                  k + "mlp.fc1.bias": (hid,), k + "mlp.fc2.weight": (d, hid), k + "mlp.fc2.bias": (d,)})
    s.update({"backbone.last_norm.weight": (d,), "backbone.last_norm.bias": (d,)})
    cin = d
    for j, cout in enumerate(spec.deconv):
        k = "keypoint_head.deconv_layers."
        s[f"{k}{3 * j}.weight"] = (cin, cout, 4, 4)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"{k}{3 * j + 1}.{n}"] = (cout,)
        cin = cout
    s["keypoint_head.final_layer.weight"] = (spec.num_joints, cin, 1, 1)
    s["keypoint_head.final_layer.bias"] = (spec.num_joints,)
    return s


def synth_params(spec: VitPoseSpec, seed: int = 0) -> dict:
    """Seeded synthetic parameters (no checkpoint exists in either environment): linear weights N(0, 1/fan_in) so that
    every sub-layer keeps O(1) activations, LayerNorm / BN affine near identity."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in vitpose_param_shapes(spec).items():
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif name.endswith("running_mean"):
            a = rng.normal(0, 0.1, shp).astype(np.float32)
        elif "norm" in name and name.endswith(".weight") or (len(shp) == 1 and name.endswith(".weight")):
            a = rng.uniform(0.8, 1.2, shp).astype(np.float32)
        elif len(shp) == 1:
            a = rng.normal(0, 0.05, shp).astype(np.float32)
        elif name.endswith("pos_embed"):
            a = (0.2 * rng.standard_normal(shp, dtype=np.float32))
        elif "deconv_layers" in name:                      # [cin][cout][4][4]: 4 taps reach each output
            a = rng.standard_normal(shp, dtype=np.float32) * np.float32(np.sqrt(2.0 / (shp[0] * 4)))
        else:
            fan_in = int(np.prod(shp[1:]))
            a = rng.standard_normal(shp, dtype=np.float32) * np.float32(np.sqrt(1.0 / fan_in))
        p[name] = a
    return p


def encoder_param_block(p: dict, spec: VitPoseSpec) -> np.ndarray:
    """flat fp32 block in the layout PP_OP_VIT_ENCODER documents (include/posepipe_hip.h)"""
    pos = p["backbone.pos_embed"][0]
    parts = [(pos[1:] + pos[:1]).astype(np.float32)]          # the cls slot's embedding is added to every token
    for i in range(spec.depth):
        k = f"backbone.blocks.{i}."
        for n in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                  "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"):
            parts.append(p[k + n])
    parts += [p["backbone.last_norm.weight"], p["backbone.last_norm.bias"]]
    return np.concatenate([np.asarray(a, np.float32).reshape(-1) for a in parts])


def build_vitpose_program(spec: VitPoseSpec, p: dict) -> Program:
    b = ProgramBuilder()
    x = b.buf(spec.in_h, spec.in_w, 4, name="input")
    gh, gw = spec.grid
    tok = b.conv(x, p["backbone.patch_embed.proj.weight"], p["backbone.patch_embed.proj.bias"], stride=spec.patch,
                 pad=spec.patch_pad, name="patch_embed")
    assert b.dims(tok) == (gh, gw, spec.dim)
    y = b.vit_encoder(tok, encoder_param_block(p, spec), depth=spec.depth, heads=spec.heads, mlp_ratio=spec.mlp_ratio)
    for j in range(len(spec.deconv)):
        k = "keypoint_head.deconv_layers."
        w = p[f"{k}{3 * j}.weight"]                               # [cin][cout][4][4]
        bn = {n: p[f"{k}{3 * j + 1}.{n}"] for n in ("weight", "bias", "running_mean", "running_var")}
        # fold BN over the OUTPUT channel axis (axis 1 of a ConvTranspose2d weight)
        wf, bf = fold_bn(np.transpose(w, (1, 0, 2, 3)), None, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"])
        if j == 0 and spec.head_bf16 and w.shape[0] % 64 == 0:
            y = b.deconv4x4s2_bf16(y, np.transpose(wf, (1, 0, 2, 3)), bf, relu=L.PP_RELU_LAST, name=f"deconv{j}.bf16")
        else:
            y = b.deconv4x4s2(y, np.transpose(wf, (1, 0, 2, 3)), bf, relu=L.PP_RELU_LAST, name=f"deconv{j}")
    hh, hw = spec.heatmap_hw
    out = b.buf(hh, hw, spec.num_joints, name="output")
    b.conv(y, p["keypoint_head.final_layer.weight"], p["keypoint_head.final_layer.bias"], out=out, out_nchw=True,
           name="final_layer")
    return b.build()
