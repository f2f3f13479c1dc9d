"""Parity at BASELINE.json's full sizes.  The CPU oracle is fast enough for a few full-size samples, and the
rest of the batch is tied to them by size-independent properties of the path: the result for a sample must
not depend on its position in the batch, on the batch size (different tile shapes / launch geometry), or on
whether the program is replayed from a hipGraph; mirroring the input mirrors the heat-map."""
import numpy as np
import pytest

from oracle import nets as onets
from posepipeline_amd import ops
from posepipeline_amd.models import hrnet, synth
from posepipeline_amd.program import Net

pytestmark = pytest.mark.gpu


def test_hrnet_w32_256x192_batch64_configs1(ctx):
    """configs[1]: HRNet-W32 256x192, batch 64 pre-cropped persons"""
    spec = hrnet.hrnet_w32_256x192()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=64)
    rng = np.random.default_rng(1)
    x = np.zeros((64, 256, 192, 4), np.float32)
    x[..., :3] = rng.standard_normal((64, 256, 192, 3)).astype(np.float32)
    hm = net.forward(x).reshape(64, 17, 64, 48)
    # full-size oracle on the first and the last sample: bit-exact
    ref = onets.HRNetRef(sd, 32).forward(np.transpose(x[[0, 63], :, :, :3], (0, 3, 1, 2)))
    assert np.array_equal(hm[[0, 63]], ref)
    # batch-position / batch-size independence: a permuted batch and small batches give the same bits
    perm = rng.permutation(64)
    assert np.array_equal(net.forward(x[perm]).reshape(64, 17, 64, 48), hm[perm])
    assert np.array_equal(net.forward(x[5:8]).reshape(3, 17, 64, 48), hm[5:8])
    assert np.array_equal(net.forward(x[40:41]).reshape(1, 17, 64, 48), hm[40:41])
    # determinism + hipGraph replay of the same program
    net.capture(64)
    assert np.array_equal(net.forward(x).reshape(64, 17, 64, 48), hm)
    # decode of the whole batch: maxvals are exactly the heat-map maxima, coordinates inside the crop window
    td = ops.TopDown(net, 17, flip_perm=hrnet.flip_perm(17), post="default")
    cs = np.tile(np.array([[96.0, 128.0, 192 / 200 * 1.25, 256 / 200 * 1.25]], np.float32), (32, 1))
    kp = td.run_precropped(x[:32], cs)
    assert kp.shape == (32, 17, 3) and np.isfinite(kp).all()
    assert (kp[:, :, 0] > -30).all() and (kp[:, :, 0] < 222).all()


def test_hrnet_w48_384x288_full_size(ctx):
    """the reference's own configuration (W48 384x288 DARK): one full-size sample against the oracle, plus the
    mirrored sample (flip_test's second pass)"""
    spec = hrnet.hrnet_w48_384x288()
    sd = synth.synth_state_dict(hrnet.hrnet_param_shapes(spec), seed=1)
    net = Net(ctx, hrnet.build_hrnet_program(spec, sd), max_batch=4)
    rng = np.random.default_rng(2)
    x = np.zeros((2, 384, 288, 4), np.float32)
    x[0, ..., :3] = rng.standard_normal((384, 288, 3)).astype(np.float32)
    x[1] = x[0][:, ::-1]
    hm = net.forward(x).reshape(2, 17, 96, 72)
    ref = onets.HRNetRef(sd, 48).forward(np.transpose(x[:, :, :, :3], (0, 3, 1, 2)))
    assert np.array_equal(hm, ref)
