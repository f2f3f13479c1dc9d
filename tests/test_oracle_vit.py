"""The ViTPose oracle (oracle/vit.py, UDP parts of oracle/preprocess.py / oracle/decode.py) has nothing in /root/reference
to be pinned on (ViTPose is absent there): PARITY UNPINNED.  What can be done on a CPU is done here: every block of
the restatement is cross-checked against an independent torch-CPU evaluation (torch.nn.functional: layer_norm, gelu,
scaled softmax attention, conv2d, conv_transpose2d), the bf16 helpers against torch.bfloat16, and the UDP transform /
DARK-UDP decode against their defining properties (exact round trip of a synthetic Gaussian through crop geometry).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import decode as odec
from oracle import preprocess as opre
from oracle import vit as OV
from posepipeline_amd.models import vitpose as MV

SPEC = MV.VitPoseSpec(dim=128, depth=2, heads=2, mlp_ratio=4, num_joints=5, deconv=(32, 32))   # head_dim 64


def test_bf16_helpers_match_torch():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4096).astype(np.float32) * 10.0 ** rng.integers(-6, 6, 4096),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.005859375, 3.3895314e38, np.inf, -np.inf], np.float32)])
    ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(OV.bf16_round(x), ref)
    assert np.array_equal(OV.bf16_from_bits(OV.bf16_bits(x)), ref)


def test_blocks_match_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((7, 192, 128), dtype=np.float32)
    g, b = rng.uniform(0.5, 1.5, 128).astype(np.float32), rng.standard_normal(128, dtype=np.float32)
    ref = F.layer_norm(torch.from_numpy(x).double(), (128,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), OV.LN_EPS)
    np.testing.assert_allclose(OV.layernorm(x, g, b), ref.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(OV.gelu(x), F.gelu(torch.from_numpy(x).double()).numpy(), rtol=1e-6, atol=1e-7)
    # attention on a packed qkv tensor, fp32 statement (no bf16 rounding) vs torch
    batch, t, heads, hd = 2, 192, 2, 64
    qkv = rng.standard_normal((batch * t, 3 * heads * hd), dtype=np.float32)
    q, k, v = (torch.from_numpy(qkv).double().reshape(batch, t, 3, heads, hd)[:, :, i].transpose(1, 2) for i in range(3))
    att = torch.softmax((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v
    ref = att.transpose(1, 2).reshape(batch * t, heads * hd).numpy()
    np.testing.assert_allclose(OV.attention(qkv, batch, t, heads, False), ref, rtol=1e-5, atol=1e-5)
    # the bf16-emulating statement stays within bf16 resolution of it
    assert np.abs(OV.attention(qkv, batch, t, heads, True) - ref).max() < 2e-2


def test_patch_embed_and_head_match_torch():
    rng = np.random.default_rng(2)
    p = MV.synth_params(SPEC, seed=7)
    x = rng.standard_normal((2, SPEC.in_h, SPEC.in_w, 4), dtype=np.float32)
    xt = torch.from_numpy(np.ascontiguousarray(np.transpose(x[..., :3], (0, 3, 1, 2)))).double()
    w, b = p["backbone.patch_embed.proj.weight"], p["backbone.patch_embed.proj.bias"]
    ref = F.conv2d(xt, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=16, padding=2)
    tok, (gh, gw) = OV.patch_embed(x, w, b)
    assert (gh, gw) == (16, 12) == tuple(ref.shape[2:])
    np.testing.assert_allclose(tok.reshape(2, gh, gw, -1), ref.permute(0, 2, 3, 1).numpy(), rtol=1e-5, atol=1e-5)
    feat = rng.standard_normal((2, gh, gw, SPEC.dim), dtype=np.float32)
    y = torch.from_numpy(np.ascontiguousarray(np.transpose(feat, (0, 3, 1, 2)))).double()
    for d, bn in ((0, 1), (3, 4)):
        k = "keypoint_head.deconv_layers."
        y = F.conv_transpose2d(y, torch.from_numpy(p[f"{k}{d}.weight"]).double(), stride=2, padding=1)
        y = F.batch_norm(y, torch.from_numpy(p[f"{k}{bn}.running_mean"]).double(), torch.from_numpy(p[f"{k}{bn}.running_var"]).double(),
                         torch.from_numpy(p[f"{k}{bn}.weight"]).double(), torch.from_numpy(p[f"{k}{bn}.bias"]).double(), False, 0.0, OV.BN_EPS)
        y = F.relu(y)
    y = F.conv2d(y, torch.from_numpy(p["keypoint_head.final_layer.weight"]).double(),
                 torch.from_numpy(p["keypoint_head.final_layer.bias"]).double())
    np.testing.assert_allclose(OV.head(feat, p), y.numpy(), rtol=1e-5, atol=1e-5)


def test_whole_model_matches_torch_fp32():
    rng = np.random.default_rng(3)
    p = MV.synth_params(SPEC, seed=9)
    x = rng.standard_normal((1, SPEC.in_h, SPEC.in_w, 4), dtype=np.float32)
    xt = torch.from_numpy(np.ascontiguousarray(np.transpose(x[..., :3], (0, 3, 1, 2)))).double()
    T = lambda a: torch.from_numpy(np.asarray(a)).double()   # noqa: E731
    h = F.conv2d(xt, T(p["backbone.patch_embed.proj.weight"]), T(p["backbone.patch_embed.proj.bias"]), stride=16, padding=2)
    gh, gw = h.shape[2:]
    h = h.flatten(2).transpose(1, 2)
    pos = T(p["backbone.pos_embed"])
    h = h + pos[:, 1:] + pos[:, :1]
    d, heads = SPEC.dim, SPEC.heads
    for i in range(SPEC.depth):
        k = f"backbone.blocks.{i}."
        y = F.layer_norm(h, (d,), T(p[k + "norm1.weight"]), T(p[k + "norm1.bias"]), OV.LN_EPS)
        qkv = F.linear(y, T(p[k + "attn.qkv.weight"]), T(p[k + "attn.qkv.bias"])).reshape(1, -1, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
        a = torch.softmax((qkv[0] * (d // heads) ** -0.5) @ qkv[1].transpose(-2, -1), -1) @ qkv[2]
        h = h + F.linear(a.transpose(1, 2).reshape(1, -1, d), T(p[k + "attn.proj.weight"]), T(p[k + "attn.proj.bias"]))
        y = F.layer_norm(h, (d,), T(p[k + "norm2.weight"]), T(p[k + "norm2.bias"]), OV.LN_EPS)
        h = h + F.linear(F.gelu(F.linear(y, T(p[k + "mlp.fc1.weight"]), T(p[k + "mlp.fc1.bias"]))), T(p[k + "mlp.fc2.weight"]), T(p[k + "mlp.fc2.bias"]))
    h = F.layer_norm(h, (d,), T(p["backbone.last_norm.weight"]), T(p["backbone.last_norm.bias"]), OV.LN_EPS)
    feat = h.reshape(1, gh, gw, d).numpy().astype(np.float32)
    ref = OV.head(feat, p)
    got = OV.forward(x, p, SPEC, emulate_bf16=False)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    got_bf = OV.forward(x, p, SPEC, emulate_bf16=True)
    assert np.abs(got_bf - ref).max() / np.abs(ref).max() < 5e-2


def test_udp_transform_and_decode_round_trip():
    """A Gaussian placed at a known IMAGE position, pushed through the UDP geometry into heat-map space, must decode
    back to that position: checks get_warp_matrix_udp, post_dark_udp and transform_preds_udp against each other."""
    bbox = np.array([310.0, 120.0, 140.0, 330.0])
    center, scale = opre.box2cs(bbox, (192, 256))
    m = opre.get_warp_matrix_udp(center, scale, (192, 256)).astype(np.float64)
    h, w = 64, 48
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    hm = np.zeros((1, 3, h, w), np.float32)
    pts = np.array([[352.25, 201.5], [400.0, 300.75], [333.3, 412.1]])
    for j, (px, py) in enumerate(pts):
        ix, iy = m[0, 0] * px + m[0, 2], m[1, 1] * py + m[1, 2]            # input-image pixel (UDP: unit = pixel spacing)
        hx, hy = ix * (w - 1) / (192 - 1), iy * (h - 1) / (256 - 1)         # heat-map coordinate
        hm[0, j] = np.exp(-((xx - hx) ** 2 + (yy - hy) ** 2) / (2 * 2.0 ** 2))
    kp, _ = odec.decode_topdown_udp(hm, None, None, center[None], scale[None], kernel=11)
    assert np.abs(kp[0, :, :2] - pts).max() < 0.05, kp[0, :, :2] - pts
