"""CPU oracle for the neural-network stages.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, op by op and unfused, the third-party forward passes the reference wrappers call:
  * mmpose 0.x `HRNet.forward` + `TopdownHeatmapSimpleHead` reached from
    pose_pipeline/wrappers/mmpose.py:75; architecture from
    3rdparty/mmpose/config/top_down/darkpose/coco/hrnet_w48_coco_384x288_dark.py:44-79
    (HRModule wiring as in SURVEY.md A3).
  * VideoPose3D `TemporalModelOptimized1f` reached from pose_pipeline/wrappers/videopose3d.py:46-85
    (strided per-window form) and `ChunkedGenerator`'s edge-replicated windows (:66-75).
mmpose / VideoPose3D are not vendored in /root/reference and not installed here, and the reference
has no tests: PARITY UNPINNED against the real third-party code.  What pins this file instead:
tests/test_oracle_nets.py checks it against an independent torch-CPU model (nn.Conv2d +
nn.BatchNorm2d in eval mode, unfolded) built from the same state_dict.

Every conv is oracle/conv_ref.c (fmaf chain over (kh, kw, cin)) on BN-folded weights; ReLU / adds /
nearest upsampling are separate numpy float32 ops in the order the reference applies them.
Activations are NHWC here; inputs/outputs are converted at the edges.
"""
from __future__ import annotations

import numpy as np

from . import clib


def fold_bn(w, gamma, beta, mean, var, eps=1e-5):
    scale = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps)
    wf = (w.astype(np.float64) * scale.reshape((-1,) + (1,) * (w.ndim - 1))).astype(np.float32)
    bf = (beta.astype(np.float64) - mean.astype(np.float64) * scale).astype(np.float32)
    return wf, bf


def relu(x):
    return np.maximum(x, np.float32(0))


class HRNetRef:
    """stages: ((modules, branches), ...) for stage2..4; widths width*2^i."""

    def __init__(self, sd, width, num_joints=17, stages=((1, 2), (4, 3), (3, 4)), blocks=4):
        self.sd, self.width, self.nj, self.stages, self.blocks = sd, width, num_joints, stages, blocks

    def cb(self, x, conv, bn, stride=1, pad=1):
        sd = self.sd
        w, b = fold_bn(sd[conv + ".weight"], sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                       sd[bn + ".running_var"])
        return clib.conv2d_nhwc(x, w, b, stride=stride, pad=pad)

    def bottleneck(self, x, p, has_ds):
        out = relu(self.cb(x, p + "conv1", p + "bn1", pad=0))
        out = relu(self.cb(out, p + "conv2", p + "bn2", pad=1))
        out = self.cb(out, p + "conv3", p + "bn3", pad=0)
        identity = self.cb(x, p + "downsample.0", p + "downsample.1", pad=0) if has_ds else x
        out = out + identity
        return relu(out)

    def basic(self, x, p):
        out = relu(self.cb(x, p + "conv1", p + "bn1"))
        out = self.cb(out, p + "conv2", p + "bn2")
        out = out + x
        return relu(out)

    def module(self, xs, mp, n_out):
        n_br = len(xs)
        xs = list(xs)
        for b in range(n_br):
            for k in range(self.blocks):
                xs[b] = self.basic(xs[b], f"{mp}branches.{b}.{k}.")
        outs = []
        for i in range(n_out):
            y = 0
            for j in range(n_br):
                f = f"{mp}fuse_layers.{i}.{j}."
                if i == j:
                    t = xs[j]
                elif j > i:
                    t = self.cb(xs[j], f + "0", f + "1", pad=0)
                    s = 2 ** (j - i)
                    t = np.repeat(np.repeat(t, s, axis=1), s, axis=2)      # nn.Upsample(mode='nearest')
                else:
                    t = xs[j]
                    for k in range(i - j):
                        t = self.cb(t, f"{f}{k}.0", f"{f}{k}.1", stride=2)
                        if k != i - j - 1:
                            t = relu(t)
                y = y + t
            outs.append(relu(y))
        return outs

    def forward(self, x_nchw):
        """x [n][3][h][w] float32 -> heatmaps [n][K][h/4][w/4]."""
        x = np.ascontiguousarray(np.transpose(np.asarray(x_nchw, np.float32), (0, 2, 3, 1)))
        B = "backbone."
        x = relu(self.cb(x, B + "conv1", B + "bn1", stride=2))
        x = relu(self.cb(x, B + "conv2", B + "bn2", stride=2))
        for i in range(4):
            x = self.bottleneck(x, f"{B}layer1.{i}.", i == 0)
        ch = [self.width * 2 ** i for i in range(4)]
        ys, pre = [x], [256]
        for si, (n_mod, n_br) in enumerate(self.stages):
            cur = ch[:n_br]
            t = f"{B}transition{si + 1}."
            xs = []
            for i in range(n_br):
                if i < len(pre):
                    xs.append(relu(self.cb(ys[i], f"{t}{i}.0", f"{t}{i}.1")) if pre[i] != cur[i] else ys[i])
                else:
                    y = ys[-1]
                    for j in range(i + 1 - len(pre)):
                        y = relu(self.cb(y, f"{t}{i}.{j}.0", f"{t}{i}.{j}.1", stride=2))
                    xs.append(y)
            for m in range(n_mod):
                last = si == len(self.stages) - 1 and m == n_mod - 1
                xs = self.module(xs, f"{B}stage{si + 2}.{m}.", 1 if last else n_br)
            ys, pre = xs, cur
        sd = self.sd
        hm = clib.conv2d_nhwc(ys[0], sd["keypoint_head.final_layer.weight"], sd["keypoint_head.final_layer.bias"])
        return np.ascontiguousarray(np.transpose(hm, (0, 3, 1, 2)))


# ---- VideoPose3D ---------------------------------------------------------------------------------
def videopose3d_windows(kp, pad=121):
    """ChunkedGenerator(chunk_length=1, pad=121): window i = frames [i-pad, i+pad], edge-replicated."""
    n = kp.shape[0]
    idx = np.clip(np.arange(-pad, pad + 1)[None, :] + np.arange(n)[:, None], 0, n - 1)
    return kp[idx]                      # [n][2*pad+1][J][2]


class VideoPose3DRef:
    """TemporalModelOptimized1f(17, 2, 17, filter_widths=[3,3,3,3,3], channels=1024), eval mode."""

    def __init__(self, sd, filter_widths=(3, 3, 3, 3, 3)):
        self.sd, self.fw = sd, tuple(filter_widths)

    def _cb(self, x, conv, bn, stride):
        sd = self.sd
        w, b = fold_bn(sd[conv + ".weight"], sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"],
                       sd[bn + ".running_var"])
        return clib.conv2d_nhwc(x, w[:, :, None, :], b, stride=stride)

    def forward(self, windows):
        """windows [B][T][J][2] -> [B][J_out][3]; strided form, one output frame per window."""
        bsz, t, j, f = windows.shape
        x = np.ascontiguousarray(windows.reshape(bsz, 1, t, j * f), dtype=np.float32)   # (B, H=1, W=T, C=34)
        x = relu(self._cb(x, "expand_conv", "expand_bn", self.fw[0]))
        for i in range(len(self.fw) - 1):
            res = x[:, :, self.fw[i + 1] // 2::self.fw[i + 1]]
            x = relu(self._cb(x, f"layers_conv.{2 * i}", f"layers_bn.{2 * i}", self.fw[i + 1]))
            x = res + relu(self._cb(x, f"layers_conv.{2 * i + 1}", f"layers_bn.{2 * i + 1}", 1))
        sd = self.sd
        y = clib.conv2d_nhwc(x, sd["shrink.weight"][:, :, None, :], sd["shrink.bias"])
        assert y.shape[2] == 1
        return y.reshape(bsz, -1, 3)
