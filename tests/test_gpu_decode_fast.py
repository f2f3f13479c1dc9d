"""flip_merge_decode_fast_kernel (round 4: register-blocked, packed-float32 separable blur on a framed, pair-interleaved LDS image;
only the blurred maximum and the 5 x 5 neighbourhood of the arg-max are kept) against the generic one-tap-at-a-time kernel it
replaces on the hot path: the SAME bits -- key points, scores, merged maps -- for every post-processing mode, on noise maps (every
tile of the map matters for the blurred maximum), peaked maps, maps with peaks on the border and all-negative maps; and against
the oracle (oracle/decode.py: mmpose's flip_back / shift / average, _gaussian_blur, _taylor, post_dark_udp, transform_preds --
pose_pipeline/utils/inference.py:27-114 is the in-tree statement of the same steps) with the suite's usual bars.
pp_debug_knob("decode_generic", 1) selects the generic kernel (process-wide; POSEPIPE_DECODE_GENERIC is read once per process)."""
import numpy as np
import pytest

from oracle import decode as odec
from posepipeline_amd import _lib as L
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet

pytestmark = pytest.mark.gpu


def _maps(rng, n, k, h, w, kind):
    if kind == "noise":
        return rng.standard_normal((n, k, h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, k, h, w), np.float32)
    for i in range(n):
        for j in range(k):
            if kind == "border":        # peaks on / next to the frame: the stencil guards and the clamped UDP stencil
                cy, cx = rng.choice([0, 1, h - 2, h - 1, h // 2]), rng.choice([0, 1, w - 2, w - 1, w // 2])
            else:
                cy, cx = rng.uniform(3, h - 4), rng.uniform(3, w - 4)
            sg = rng.uniform(1.5, 3.5)
            out[i, j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg)) + 0.01 * rng.random((h, w))
    if kind == "negative":
        out = -np.abs(out) - 0.1
    return out


CASES = [(96, 72, "unbiased", 17), (64, 48, "unbiased", 11), (64, 48, "udp", 11), (96, 72, "udp", 17), (96, 72, "default", 17),
         (64, 48, None, 11), (24, 16, "unbiased", 17), (12, 8, "udp", 11), (128, 128, "unbiased", 17), (10, 12, "unbiased", 11),
         # 8-pixel maps with the 17-tap kernel under UDP's reflected frame: the launcher hands them to the generic kernel
         (8, 8, "udp", 17), (8, 12, "udp", 17), (8, 8, "unbiased", 17)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["noise", "peaked", "border", "negative"])
def test_fast_decode_equals_generic_kernel(ctx, monkeypatch, case, kind):
    h, w, post, ks = case
    rng = np.random.default_rng(h * 131 + w + ks + len(kind))
    n, k = 3, 17
    hm, hf = _maps(rng, n, k, h, w, kind), _maps(rng, n, k, h, w, kind)
    cs = np.concatenate([rng.uniform(100, 900, (n, 2)), rng.uniform(0.8, 3.0, (n, 2))], axis=1).astype(np.float32)
    perm = hrnet.flip_perm(17)
    for flip, shift in ((True, True), (True, False), (False, False)):
        args = dict(flip_perm=perm if flip else None, shift_heatmap=shift, post=post, blur_kernel=ks, want_merged=True)
        try:
            L.check(ctx.lib.pp_debug_knob(b"decode_generic", 1), "pp_debug_knob")
            kp_ref, mg_ref = ops.flip_merge_decode(ctx, hm, hf if flip else None, cs, **args)
        finally:
            L.check(ctx.lib.pp_debug_knob(b"decode_generic", -1), "pp_debug_knob")
        kp, mg = ops.flip_merge_decode(ctx, hm, hf if flip else None, cs, **args)
        assert np.array_equal(mg, mg_ref)
        assert np.array_equal(kp, kp_ref, equal_nan=True), (case, kind, flip, shift, np.abs(kp - kp_ref).max())


@pytest.mark.parametrize("post,ks,size", [("unbiased", 17, (96, 72)), ("unbiased", 11, (64, 48)), ("default", 17, (64, 48))])
def test_fast_decode_against_the_oracle(ctx, post, ks, size):
    h, w = size
    rng = np.random.default_rng(7)
    n = 4
    hm, hf = _maps(rng, n, 17, h, w, "peaked"), _maps(rng, n, 17, h, w, "peaked")
    cs = np.concatenate([rng.uniform(100, 900, (n, 2)), rng.uniform(0.8, 3.0, (n, 2))], axis=1).astype(np.float32)
    kp, merged = ops.flip_merge_decode(ctx, hm, hf, cs, flip_perm=hrnet.flip_perm(17), post=post, blur_kernel=ks, want_merged=True)
    ref, _ = odec.decode_topdown(hm, hf, hrnet.COCO_FLIP_PAIRS, cs[:, :2], cs[:, 2:], post_process=post, kernel=ks)
    assert np.array_equal(kp[:, :, 2], ref[:, :, 2].astype(np.float32))                  # scores: the merged maximum, bit for bit
    assert np.abs(kp[:, :, :2] - ref[:, :, :2]).max() <= 1e-3                            # px (numpy float32 log vs the device's rounded one)
