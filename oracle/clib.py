"""ctypes binding of oracle/liboracle.so (oracle/conv_ref.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)


def usable_cores() -> int:
    """min(scheduler affinity, cgroup CPU quota): what a process here can actually run in parallel."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


N_THREADS = None


def lib():
    global _lib, N_THREADS
    if _lib is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:
            build(force=True)
            _lib = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        _lib.oracle_conv2d_nhwc.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp] + [C.c_int] * 9 + [fp]
        _lib.oracle_conv2d_nhwc.restype = None
        _lib.oracle_maxpool2d_nhwc.argtypes = [fp] + [C.c_int] * 9 + [fp]
        _lib.oracle_maxpool2d_nhwc.restype = None
        N_THREADS = usable_cores()
        _lib.oracle_set_threads.argtypes = [C.c_int]
        _lib.oracle_set_threads(N_THREADS)
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def out_dim(i, k, s, p, d=1):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def conv2d_nhwc(x, weight, bias=None, stride=1, pad=(0, 0), dil=(1, 1)):
    """x [n][h][w][cin] fp32; weight torch layout [cout][cin][kh][kw]; fmaf chain over (kh, kw, cin)."""
    if isinstance(pad, int):
        pad = (pad, pad)
    if isinstance(dil, int):
        dil = (dil, dil)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.asarray(weight, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    n, h, wd, cx = x.shape
    assert cx == cin, (cx, cin)
    wk = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)).reshape(kh * kw * cin, cout))
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    ho, wo = out_dim(h, kh, stride, pad[0], dil[0]), out_dim(wd, kw, stride, pad[1], dil[1])
    y = np.empty((n, ho, wo, cout), dtype=np.float32)
    lib().oracle_conv2d_nhwc(_fp(x), n, h, wd, cin, _fp(wk), _fp(b) if b is not None else None, cout, cout, kh, kw,
                             stride, pad[0], pad[1], dil[0], dil[1], _fp(y))
    return y


def maxpool2d_nhwc(x, k, stride, pad):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, h, w, c = x.shape
    y = np.empty((n, out_dim(h, k, stride, pad), out_dim(w, k, stride, pad), c), dtype=np.float32)
    lib().oracle_maxpool2d_nhwc(_fp(x), n, h, w, c, k, k, stride, pad, pad, _fp(y))
    return y
