"""HIP crop/normalise, HRNet forward and flip-merge + DARK decode vs the CPU oracle."""
import numpy as np
import pytest

from oracle import decode as odec
from oracle import nets as onets
from oracle import preprocess as opre
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.program import Net

pytestmark = pytest.mark.gpu


def synth_frames(rng, n, h, w):
    # low-frequency background + texture so that bilinear weights matter
    base = rng.integers(0, 256, (n, h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :h, :w]
    noise = rng.integers(-20, 21, fr.shape)
    return np.clip(fr.astype(np.int64) + noise, 0, 255).astype(np.uint8)


def test_crop_affine_normalize_bit_exact(ctx):
    rng = np.random.default_rng(11)
    frames = synth_frames(rng, 3, 270, 480)
    bboxes = np.array([
        [100.3, 40.7, 80.2, 190.9],      # inside
        [-30.5, -20.25, 200.0, 150.0],   # hangs over the top-left corner -> zero border taps
        [400.0, 200.0, 120.0, 100.0],    # over the bottom-right corner, wide box (aspect fix on h)
        [np.nan, np.nan, np.nan, np.nan],  # absent person (wrappers/mmpose.py:67-69)
        [10.0, 10.0, 30.0, 60.0],        # small box: magnification
        [0.0, 0.0, 480.0, 270.0],        # whole frame: minification
    ], dtype=np.float64)
    fidx = np.array([0, 1, 2, 0, 1, 2], dtype=np.int32)
    # the wrapper feeds cv2's BGR frame swapped to RGB; mmpose swaps again -> tensor channel c = BGR channel c
    res = ops.crop_affine_normalize(ctx, frames, fidx, bboxes, out_wh=(288, 384), chan_map=(0, 1, 2), flip=True,
                                    want_crop_u8=True)
    out = res["out"]
    for i, bb in enumerate(bboxes):
        if np.isnan(bb).any():
            assert res["valid"][i] == 0 and not out[i].any() and not out[len(bboxes) + i].any()
            continue
        wrapper_rgb = frames[fidx[i]][:, :, ::-1]
        t, c, s, crop = opre.top_down_input(wrapper_rgb, bb, (288, 384))
        assert res["valid"][i] == 1
        assert np.array_equal(res["center_scale"][i], np.concatenate([c, s]))
        assert np.array_equal(res["crop_u8"][i], crop), f"person {i}: {np.abs(res['crop_u8'][i].astype(int) - crop).max()}"
        got = np.transpose(out[i][:, :, :3], (2, 0, 1))
        assert np.array_equal(got, t)
        assert not out[i][:, :, 3].any()
        assert np.array_equal(out[len(bboxes) + i], out[i][:, ::-1])     # img.flip(3)


@pytest.mark.parametrize("post", ["unbiased", "default"])
def test_flip_merge_decode(ctx, post):
    rng = np.random.default_rng(5)
    n, k, h, w = 3, 17, 96, 72
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.zeros((n, k, h, w), np.float32)
    hmf = np.zeros_like(hm)
    perm = hrnet.flip_perm(k)
    for i in range(n):
        for j in range(k):
            cx, cy = rng.uniform(0, w - 1), rng.uniform(0, h - 1)
            if j == 3:
                cx, cy = 0.4, 1.2             # border: Taylor step must be skipped
            if j == 4:
                cx, cy = w - 2.6, h - 1.1
            g = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 3.0 ** 2)).astype(np.float32)
            hm[i, j] = g * rng.uniform(0.3, 1.0) + rng.normal(0, 0.01, (h, w))
            gf = np.exp(-((xx - (w - 1 - cx) - 1) ** 2 + (yy - cy) ** 2) / (2 * 3.0 ** 2)).astype(np.float32)
            hmf[i, perm[j]] = gf * rng.uniform(0.3, 1.0) + rng.normal(0, 0.01, (h, w))
    hm[1, 7] = -np.abs(hm[1, 7]) - 1e-3     # no positive peak -> preds = -1
    hmf[1, perm[7]] = -np.abs(hmf[1, perm[7]]) - 1e-3
    hm[2, 9] = 0.0
    hmf[2, perm[9]] = 0.0                   # all-zero map: argmax 0, maxval 0 -> -1
    center = rng.uniform(100, 900, (n, 2)).astype(np.float32)
    scale = rng.uniform(0.5, 4.0, (n, 2)).astype(np.float32)
    ref, ref_merged = odec.decode_topdown(hm, hmf, hrnet.COCO_FLIP_PAIRS, center, scale, post_process=post, kernel=17)
    got, merged = ops.flip_merge_decode(ctx, hm, hmf, np.concatenate([center, scale], 1), flip_perm=perm, post=post,
                                        want_merged=True)
    assert np.array_equal(merged, ref_merged)
    assert np.array_equal(got[:, :, 2], ref[:, :, 2])                         # maxvals: bit-exact
    # coordinates: north_star tolerance 1e-3 px (float32 log differs by <= 1 ulp between numpy and the GPU)
    assert np.abs(got[:, :, :2] - ref[:, :, :2]).max() <= 1e-3, np.abs(got[:, :, :2] - ref[:, :, :2]).max()


def test_hrnet_w32_forward_bit_exact(ctx):
    spec = hrnet.HRNetSpec(32, 17, 96, 64)       # same network, small input so the CPU oracle takes seconds
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    prog = hrnet.build_hrnet_program(spec, sd)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((3, 3, 96, 64)).astype(np.float32)
    ref = onets.HRNetRef(sd, 32).forward(x)
    net = Net(ctx, prog, max_batch=4)
    xin = np.zeros((3, 96, 64, 4), np.float32)
    xin[..., :3] = np.transpose(x, (0, 2, 3, 1))
    got = net.forward(xin)                        # named output is NCHW planes [n][17][24][16]
    got = got.reshape(3, 17, 24, 16)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1e-3
    assert np.array_equal(got, ref), np.abs(got - ref).max()


def test_hrnet_fuse_layers_one_pass_form_is_bit_identical(ctx, monkeypatch):
    """PP_OP_UPSAMPLE_ADD: the fuse layers as coarse 1x1 convs + ONE upsample-accumulate pass per output (up to three coarse
    terms, mmpose's summation order) give the same bits as round 1's form, in which every 1x1 conv's epilogue scatters
    (partial + value) over its 2^u x 2^u patch -- and both equal the oracle (the full-size tests)"""
    spec = hrnet.HRNetSpec(32, 17, 96, 64)
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    rng = np.random.default_rng(2)
    x = np.zeros((3, 96, 64, 4), np.float32)
    x[..., :3] = rng.standard_normal((3, 96, 64, 3)).astype(np.float32)
    outs = []
    for mode in ("conv", "onepass"):
        monkeypatch.setattr(hrnet, "FUSE_MODE", mode)
        prog = hrnet.build_hrnet_program(spec, sd)
        net = Net(ctx, prog, 3)
        outs.append((len(prog.ops), sum(1 for o in prog.ops if o.type == 7 and o.in3 >= 0), net.forward(x)))
        net.close()
    assert outs[1][0] > outs[0][0] and outs[0][1] == 0 and outs[1][1] >= 1     # the one-pass program has three-term upsample_add ops
    assert np.array_equal(outs[0][2], outs[1][2])
