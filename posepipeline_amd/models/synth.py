"""Seeded synthetic parameters (no checkpoints exist in either environment).

He-normal conv weights; BatchNorm gamma ~ U(.5,1), beta ~ N(0,.1), running_mean ~ N(0,.1),
running_var ~ U(.5,1.5).  The last BN of every residual block gets a small gamma (U(.1,.3)) so that
activations keep O(1) scale through ~36 residual blocks instead of doubling their variance each time.
"""
from __future__ import annotations

import re

import numpy as np

_LAST_BN = re.compile(r"(layer\d+\.\d+\.bn3|branches\.\d+\.\d+\.bn2|layer\d+\.\d+\.bn2$|layers_bn\.(1|3|5|7))")


def synth_state_dict(shapes: dict, seed: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shp in shapes.items():
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shp)
        elif name.endswith("running_mean"):
            a = rng.normal(0, 0.1, shp)
        elif len(shp) == 1 and name.endswith(".weight"):            # BN gamma
            stem = name[: -len(".weight")]
            a = rng.uniform(0.1, 0.3, shp) if _LAST_BN.search(stem) else rng.uniform(0.5, 1.0, shp)
        elif len(shp) == 1:                                         # BN beta / conv bias
            a = rng.normal(0, 0.1, shp)
        else:                                                       # conv / linear weight
            fan_in = int(np.prod(shp[1:]))
            a = rng.normal(0, np.sqrt(2.0 / fan_in), shp)
        sd[name] = a.astype(np.float32)
    return sd


_LAST_CONV = re.compile(r"(layer\d+\.\d+\.conv3|branches\.\d+\.\d+\.conv2)\.weight$")
_UP_FUSE = re.compile(r"fuse_layers\.(\d+)\.(\d+)\.0\.weight$")


def smooth_state_dict(shapes: dict, seed: int = 0, centre: float = 30.0, up_gain: float = 0.05, diag: float = 2.0) -> dict:
    """WELL-CONDITIONED parameters for the pose networks (parity tests of the default numerics at full size).

    Seeded-random weights make heat-maps noise: arg-maxes sit between near-equal maxima and DARK's Newton step divides by
    near-singular Hessians, so ANY float32 reordering moves some joints by more than 1e-3 px.  A trained
    network produces smooth single-peaked maps.  This constructor gives the same property without a checkpoint: every
    convolution kernel is positive and normalised (a smoothing / averaging kernel, centre tap heavier so that ~70 layers
    do not blur a blob away; where cin == cout most of the mass stays on the same channel, so channels keep their own blob
    position), BatchNorm is the identity, the second branch of every residual block and the cross-resolution 1x1 fuse layers
    are damped, and each joint of the head reads a few feature channels.  On blob images the heat-maps
    are then smooth, positive, single-peaked -- and every float32 evaluation order decodes to the same joint within ~1e-4 px."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shp in shapes.items():
        if name.endswith("running_var"):
            a = np.ones(shp)
        elif name.endswith("running_mean"):
            a = np.zeros(shp)
        elif len(shp) == 1 and name.endswith(".weight"):            # BN gamma
            a = np.ones(shp)
        elif len(shp) == 1:                                         # BN beta / conv bias
            a = np.zeros(shp)
        elif name.startswith("keypoint_head"):                      # [K][C][1][1]: joint j follows a few channels
            a = np.zeros(shp)
            for j in range(shp[0]):
                a[j, rng.choice(shp[1], 3, replace=False), 0, 0] = (1.0, 0.4, 0.15)
        else:
            a = np.abs(rng.normal(0, 1, shp)) + 0.05
            if len(shp) == 4 and shp[2] == 3 and shp[3] == 3:
                a[:, :, 1, 1] *= centre
            if len(shp) == 4 and shp[0] == shp[1]:                  # keep channel identity: ~2/3 of the mass on the same channel
                idx = np.arange(shp[0])
                a[idx, idx] *= diag * shp[1]
            gain = 1.0
            if _LAST_CONV.search(name):
                gain = 0.25
            m = _UP_FUSE.search(name)
            if m and int(m.group(2)) > int(m.group(1)):
                gain = up_gain
            a *= gain / a.reshape(shp[0], -1).sum(axis=1).reshape((-1,) + (1,) * (len(shp) - 1))
        sd[name] = a.astype(np.float32)
    return sd


def smooth_lifting_state_dict(shapes: dict, seed: int = 0, residual_gain: float = 0.25, shrink_gain: float = 0.4) -> dict:
    """WELL-CONDITIONED parameters for VideoPose3D (`videopose3d_param_shapes`): the lifting analogue of `smooth_state_dict`.

    A trained lifting network maps screen-normalised 2D joints (about [-1, 1]) to root-relative joints in METRES (about
    [-1, 1] again) with a sensitivity of order one.  Seeded He-normal weights have neither property by construction (measured:
    outputs up to 1.9, max-norm sensitivity 5.5), so a tolerance stated in millimetres says little on them.  Here both hold BY
    CONSTRUCTION: every layer is a contraction in the max norm with a known row sum --
      expand_conv        3 signed entries per channel, |row| = 1, beta = 0.3 keeps most channels above the ReLU threshold;
      layers_conv.2i     positive, two thirds of each row on the same channel's centre tap, row sum 1 (a temporal smoother);
      layers_conv.2i+1   positive, diagonal-heavy, row sum `residual_gain` (damped residual branch: x + relu(.) grows by at most
                         1 + residual_gain per block);
      shrink             4 signed entries per output coordinate, |row| = shrink_gain, bias = a skeleton-sized offset in metres;
      BatchNorm          identity
    so |d out|_max <= 1 * (1 + residual_gain)^4 * shrink_gain * |d in|_max = 0.98 |d in|_max for the defaults: an error of
    1e-3 px in a 1920-px-wide frame (1.04e-6 normalised) cannot become more than 1.02e-6 m, and outputs span about a metre."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shp in shapes.items():
        if name.endswith("running_var"):
            a = np.ones(shp)
        elif name.endswith("running_mean"):
            a = np.zeros(shp)
        elif name == "expand_bn.bias":
            a = np.full(shp, 0.3)
        elif name == "shrink.bias":
            a = rng.normal(0, 0.25, shp)
        elif len(shp) == 1 and name.endswith(".weight"):            # BN gamma
            a = np.ones(shp)
        elif len(shp) == 1:                                         # BN beta
            a = np.zeros(shp)
        elif name in ("expand_conv.weight", "shrink.weight"):       # sparse signed rows
            a = np.zeros(shp)
            flat = a.reshape(shp[0], -1)
            vals = np.array((0.6, 0.3, 0.1)) if name.startswith("expand") else np.array((0.4, 0.3, 0.2, 0.1)) * shrink_gain
            for r in range(shp[0]):
                idx = rng.choice(flat.shape[1], len(vals), replace=False)
                flat[r, idx] = vals * rng.choice((-1.0, 1.0), len(vals))
        else:                                                       # [C][C][k] temporal convolutions of the residual blocks
            a = np.abs(rng.normal(0, 1, shp)) + 0.05
            idx = np.arange(shp[0])
            a[idx, idx, shp[2] // 2] *= 2.0 * shp[1] * shp[2]
            gain = residual_gain if shp[2] == 1 else 1.0
            a *= gain / a.reshape(shp[0], -1).sum(axis=1)[:, None, None]
        sd[name] = a.astype(np.float32)
    return sd


def max_norm_gain_bound(sd: dict) -> float:
    """upper bound of |d out|_max / |d in|_max of a VideoPose3D state dict with identity BatchNorm: product over the layers of the
    largest absolute row sum, residual blocks as 1 + (row sum of the first conv) * (row sum of the second)"""
    row = lambda k: float(np.abs(sd[k].astype(np.float64)).reshape(sd[k].shape[0], -1).sum(axis=1).max())
    g = row("expand_conv.weight")
    i = 0
    while f"layers_conv.{2 * i}.weight" in sd:
        g *= 1.0 + row(f"layers_conv.{2 * i}.weight") * row(f"layers_conv.{2 * i + 1}.weight")
        i += 1
    return g * row("shrink.weight")


def blob_crops(rng, n, h, w, channels=3, sigma=(6.0, 10.0)):
    """[n][h][w][4] float32 network inputs (4th channel 0): one Gaussian blob per sample, the colour planes slightly apart
    (what a normalised crop of a bright person on a dark background looks like to a smoothing network)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    x = np.zeros((n, h, w, 4), np.float32)
    for i in range(n):
        cy, cx = rng.uniform(0.3 * h, 0.7 * h), rng.uniform(0.3 * w, 0.7 * w)
        for c in range(channels):
            sg = rng.uniform(*sigma)
            oy, ox = rng.uniform(-6, 6, 2)
            x[i, :, :, c] = rng.uniform(0.6, 1.4) * np.exp(-((yy - cy - oy) ** 2 + (xx - cx - ox) ** 2) / (2 * sg * sg))
    x[..., :channels] += rng.uniform(0, 1e-3, (n, h, w, channels)).astype(np.float32)
    return x
