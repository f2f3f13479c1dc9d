"""CPU oracle for the ViTPose backbone + head (BASELINE.json configs[4]).  TEST INFRASTRUCTURE ONLY.

ViTPose is NOT in the reference tree ("C5: ViTPose-H is not in the reference -> out of contract"); it would
be loaded through the same calls as the HRNet models (pose_pipeline/wrappers/mmpose.py:57 `init_pose_model`, :75
`inference_top_down_pose_model`).  The architecture is restated from the published ViTPose code (mmpose 0.x fork:
`ViT` backbone with PatchEmbed(patch 16, padding 4 + 2 * (ratio // 2 - 1) = 2), pos_embed[:, 1:] + pos_embed[:, :1],
pre-norm blocks, LayerNorm eps 1e-6, `last_norm`; `TopdownHeatmapSimpleHead` with two ConvTranspose2d(4, 2, 1) + BN +
ReLU and a 1x1 conv) and is PARITY UNPINNED: there is nothing in /root/reference to pin it on.

Two statements of the same network:
  * `forward(..., emulate_bf16=True)`  -- rounds the operands of every contraction to bfloat16 (RNE) at exactly the
    points the HIP path does (include/posepipe_hip.h, "ViT encoder") and accumulates in float64: the closest CPU
    model of the bf16 MFMA path; the GPU differs from it only by fp32 accumulation order.
  * `forward(..., emulate_bf16=False)` -- plain float64/float32 arithmetic: what the published fp32 model computes.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf

F32 = np.float32
LN_EPS = 1e-6
BN_EPS = 1e-5


def bf16_round(x):
    """float32 -> float32 holding the nearest bfloat16 (ties to even)."""
    u = np.ascontiguousarray(x, F32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(F32)


def bf16_bits(x):
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b):
    return (np.asarray(b, np.uint16).astype(np.uint32) << 16).view(F32)


def _q(x, on):
    return bf16_round(x) if on else np.asarray(x, F32)


def linear(x, w, b, q):
    """x [m][k], w [n][k] (nn.Linear), float64 accumulation of (optionally bf16-rounded) operands -> float32"""
    y = _q(x, q).astype(np.float64) @ _q(w, q).astype(np.float64).T
    if b is not None:
        y = y + b.astype(np.float64)
    return y.astype(F32)


def layernorm(x, g, b, eps=LN_EPS):
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return ((x64 - mu) / np.sqrt(var + eps) * g.astype(np.float64) + b.astype(np.float64)).astype(F32)


def gelu(x):
    x64 = x.astype(np.float64)
    return (0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))).astype(F32)


def attention(qkv, batch, tokens, heads, q):
    """qkv [batch * tokens][3 * dim] with columns (q|k|v, head, head_dim) -> [batch * tokens][dim]"""
    dim = qkv.shape[1] // 3
    hd = dim // heads
    x = _q(qkv, q).reshape(batch, tokens, 3, heads, hd).astype(np.float64)
    qq, kk, vv = (np.transpose(x[:, :, i], (0, 2, 1, 3)) for i in range(3))      # [b][h][t][hd]
    s = (qq @ np.transpose(kk, (0, 1, 3, 2))).astype(F32).astype(np.float64)    # fp32 scores
    e = np.exp((s - s.max(-1, keepdims=True)) * (1.0 / np.sqrt(np.float32(hd)).astype(np.float64)))
    den = e.sum(-1, keepdims=True)
    eq = _q(e.astype(F32), q).astype(np.float64)                                 # numerators are the MFMA operand
    o = (eq @ vv) / den
    return np.transpose(o, (0, 2, 1, 3)).reshape(batch * tokens, dim).astype(F32)


def encoder(tokens_in, p, spec, q=True):
    """tokens_in [b][t][dim] fp32 patch embeddings -> [b][t][dim] after last_norm"""
    b, t, d = tokens_in.shape
    pos = p["backbone.pos_embed"][0]
    x = (tokens_in + (pos[1:] + pos[:1])[None]).astype(F32).reshape(b * t, d)
    for i in range(spec.depth):
        k = f"backbone.blocks.{i}."
        h = layernorm(x, p[k + "norm1.weight"], p[k + "norm1.bias"])
        qkv = linear(h, p[k + "attn.qkv.weight"], p[k + "attn.qkv.bias"], q)
        a = attention(qkv, b, t, spec.heads, q)
        x = (x + linear(a, p[k + "attn.proj.weight"], p[k + "attn.proj.bias"], q)).astype(F32)
        h = layernorm(x, p[k + "norm2.weight"], p[k + "norm2.bias"])
        m = gelu(linear(h, p[k + "mlp.fc1.weight"], p[k + "mlp.fc1.bias"], q))
        x = (x + linear(m, p[k + "mlp.fc2.weight"], p[k + "mlp.fc2.bias"], q)).astype(F32)
    x = layernorm(x, p["backbone.last_norm.weight"], p["backbone.last_norm.bias"])
    return x.reshape(b, t, d)


def patch_embed(x_nhwc, w, bias, patch=16, pad=2):
    """Conv2d(3, dim, 16, stride 16, padding 2) on NHWC input (channels beyond the weight's are ignored) -> [b][t][dim]"""
    b, hh, ww, _ = x_nhwc.shape
    cin = w.shape[1]
    xp = np.zeros((b, hh + 2 * pad, ww + 2 * pad, cin), np.float64)
    xp[:, pad:pad + hh, pad:pad + ww] = x_nhwc[..., :cin]
    gh, gw = (hh + 2 * pad - patch) // patch + 1, (ww + 2 * pad - patch) // patch + 1
    xp = xp[:, :gh * patch, :gw * patch].reshape(b, gh, patch, gw, patch, cin)
    cols = np.transpose(xp, (0, 1, 3, 5, 2, 4)).reshape(b * gh * gw, cin * patch * patch)     # (c, kh, kw)
    y = cols @ w.reshape(w.shape[0], -1).astype(np.float64).T + bias.astype(np.float64)
    return y.astype(F32).reshape(b, gh * gw, -1), (gh, gw)


def conv_transpose_4s2p1(x_nhwc, w):
    """ConvTranspose2d(kernel 4, stride 2, padding 1, bias False); w torch layout [cin][cout][4][4]; float64."""
    b, h, ww, cin = x_nhwc.shape
    cout = w.shape[1]
    full = np.zeros((b, 2 * h + 2, 2 * ww + 2, cout), np.float64)
    x64 = x_nhwc.astype(np.float64).reshape(b * h * ww, cin)
    w64 = np.ascontiguousarray(np.transpose(w.astype(np.float64), (2, 3, 0, 1)))      # [kh][kw][cin][cout]
    for kh in range(4):
        for kw in range(4):
            full[:, kh:kh + 2 * h:2, kw:kw + 2 * ww:2] += (x64 @ w64[kh, kw]).reshape(b, h, ww, cout)
    return full[:, 1:1 + 2 * h, 1:1 + 2 * ww]


def head(feat_nhwc, p, q_first=False):
    """keypoint_head: deconv_layers (0 deconv, 1 BN, 2 ReLU, 3 deconv, 4 BN, 5 ReLU) + final_layer 1x1 -> NCHW heatmaps.
    q_first: the HIP path runs the FIRST deconvolution on the bf16 matrix cores (PP_OP_DECONV_BF16): BatchNorm folded into
    the weights (float64, one rounding to float32, as posepipeline_amd.program.fold_bn), input and folded weights rounded
    to bf16, then the folded bias."""
    x = feat_nhwc
    for d, bn in ((0, 1), (3, 4)):
        k = "keypoint_head.deconv_layers."
        g, be = p[f"{k}{bn}.weight"].astype(np.float64), p[f"{k}{bn}.bias"].astype(np.float64)
        mu, var = p[f"{k}{bn}.running_mean"].astype(np.float64), p[f"{k}{bn}.running_var"].astype(np.float64)
        if d == 0 and q_first:
            scale = g / np.sqrt(var + BN_EPS)
            wf = (p[f"{k}{d}.weight"].astype(np.float64) * scale[None, :, None, None]).astype(F32)
            bf = (be - mu * scale).astype(F32)
            y = conv_transpose_4s2p1(bf16_round(x), bf16_round(wf)) + bf.astype(np.float64)
            x = np.maximum(y, 0.0).astype(F32)
            continue
        y = conv_transpose_4s2p1(x, p[f"{k}{d}.weight"])
        x = np.maximum((y - mu) / np.sqrt(var + BN_EPS) * g + be, 0.0).astype(F32)
    wf = p["keypoint_head.final_layer.weight"][:, :, 0, 0].astype(np.float64)
    y = x.astype(np.float64) @ wf.T + p["keypoint_head.final_layer.bias"].astype(np.float64)
    return np.ascontiguousarray(np.transpose(y, (0, 3, 1, 2))).astype(F32)


def forward(x_nhwc, p, spec, emulate_bf16=True):
    """x [b][256][192][>=3] normalised image -> heatmaps [b][K][64][48]"""
    tok, (gh, gw) = patch_embed(x_nhwc, p["backbone.patch_embed.proj.weight"], p["backbone.patch_embed.proj.bias"])
    y = encoder(tok, p, spec, emulate_bf16)
    return head(y.reshape(y.shape[0], gh, gw, -1), p, q_first=emulate_bf16 and getattr(spec, "head_bf16", False))
