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
