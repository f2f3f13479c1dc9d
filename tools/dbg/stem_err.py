"""error statistics of the stem kernel vs the float32 kernel (debug aid): python tools/dbg/stem_err.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from posepipeline_amd import _lib as L  # noqa: E402
from tests.helpers import hip_conv_op  # noqa: E402
from tests.test_gpu_split import conv64  # noqa: E402

ctx = L.Context(0)
L.check(ctx.lib.pp_conv_split_kind(1))
for (n, h, w) in [(2, 64, 128), (5, 270, 480)]:
    rng = np.random.default_rng(h * w)
    wt = (rng.standard_normal((64, 4, 7, 7)) / np.sqrt(147)).astype(np.float32)
    wt[:, 3] = 0
    wt *= np.exp(2 * rng.standard_normal((64, 1, 1, 1))).astype(np.float32)
    for dist in ("lognormal", "normal", "image"):
        if dist == "lognormal":
            x = (rng.standard_normal((n, h, w, 4)) * np.exp(rng.standard_normal((n, h, w, 4)))).astype(np.float32)
        elif dist == "normal":
            x = rng.standard_normal((n, h, w, 4)).astype(np.float32)
        else:
            x = ((rng.integers(0, 256, (n, h, w, 4)) - 110.0) / 58.0).astype(np.float32)
        x[..., 3] = 0
        b = rng.standard_normal(64).astype(np.float32)
        ref = conv64(x, wt, b, 3, 2)
        out = []
        for exact in (1, 0):
            L.check(ctx.lib.pp_conv_exact(exact))
            out.append(hip_conv_op(ctx, x, wt, b, stride=2, pad=(3, 3)))
        scale = np.abs(ref).reshape(n, -1, 64).max(1).reshape(n, 1, 1, 64)
        e = [np.abs((o - ref) / scale) for o in out]
        print(n, h, w, dist, "rms exact %.3e split %.3e ratio %.2f | max exact %.3e split %.3e ratio %.2f" % (
            np.sqrt((e[0] ** 2).mean()), np.sqrt((e[1] ** 2).mean()), np.sqrt((e[1] ** 2).mean()) / np.sqrt((e[0] ** 2).mean()),
            e[0].max(), e[1].max(), e[1].max() / e[0].max()), flush=True)
