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
