import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """A HIP context on device 0.  GPU tests must fail (not skip) when the library or the GPU is missing."""
    from posepipeline_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _collect_device_objects(request):
    """nets / detectors / cascades free their device memory in __del__; cascades are reference cycles, so a full suite in one process
    kept tens of GB alive between collections (round 6: the 1080p parity tests pushed a later test out of memory).  Collect after
    every GPU test."""
    yield
    if request.node.get_closest_marker("gpu"):
        import gc
        gc.collect()


@pytest.fixture(autouse=True)
def _conv_numerics(request):
    """GPU tests create their nets with the float32-MFMA convolution kernels, which are bit-identical to oracle/conv_ref.c, so
    that `==` against the oracle is meaningful.  A net keeps the numerics it was created with (ABI 7).  The library's DEFAULT
    (three-way bf16 split on the bf16 matrix cores: float32-accurate but not bit-identical) is covered by
    tests/test_gpu_parity_modes.py (the oracle-comparing end-to-end tests in BOTH modes, north_star tolerances) and
    tests/test_gpu_split.py (per layer / per network against the bit-exact kernels)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from posepipeline_amd import _lib
    # POSEPIPE_TEST_NUMERICS=default: run the suite on the default kernels instead (the `==` assertions then fail by design; used
    # to check that nothing else -- shapes, ids, error paths, graph capture, streaming -- depends on the numerics mode)
    mode = "split" if os.environ.get("POSEPIPE_TEST_NUMERICS") == "default" else "exact"
    old = os.environ.get("POSEPIPE_CONV_EXACT")
    os.environ["POSEPIPE_CONV_EXACT"] = "1" if mode == "exact" else "0"     # worker processes a test starts (sharded ranks, env-knob probes)
    with _lib.default_numerics(mode):
        yield
    if old is None:
        os.environ.pop("POSEPIPE_CONV_EXACT", None)
    else:
        os.environ["POSEPIPE_CONV_EXACT"] = old
