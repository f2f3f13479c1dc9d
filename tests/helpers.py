"""Shared helpers for the parity tests (oracle-side composition of one fused conv op)."""
import ctypes as C

import numpy as np

from oracle import clib
from posepipeline_amd import _lib as L
from posepipeline_amd.program import pack_conv


def ref_conv_op(x, weight, bias, *, stride=1, pad=(0, 0), dil=(1, 1), relu=0, res1=None, res2=None, up_log2=0,
                out_nchw=False, res1_shift=0, res1_off_w=0):
    """Oracle statement of one pp_op: conv (fmaf chain) + bias, then the documented epilogue order."""
    y = clib.conv2d_nhwc(x, weight, bias if bias is not None else np.zeros(weight.shape[0], np.float32), stride=stride,
                         pad=pad, dil=dil)
    if relu == L.PP_RELU_FIRST:
        y = np.maximum(y, np.float32(0))
    f = 1 << up_log2
    if f > 1:
        y = np.repeat(np.repeat(y, f, axis=1), f, axis=2)
    if res1 is not None:
        r = res1
        if res1_shift:
            s = 1 << res1_shift
            r = np.repeat(np.repeat(r, s, axis=1), s, axis=2)[:, : y.shape[1], : y.shape[2]]
        if res1_off_w:
            r = r[:, :, res1_off_w: res1_off_w + y.shape[2]]
        y = (y + r).astype(np.float32)
    if res2 is not None:
        y = (y + res2).astype(np.float32)
    if relu == L.PP_RELU_LAST:
        y = np.maximum(y, np.float32(0))
    if out_nchw:
        y = np.ascontiguousarray(np.transpose(y, (0, 3, 1, 2)))
    return y


def hip_conv_op(ctx, x, weight, bias, *, stride=1, pad=(0, 0), dil=(1, 1), relu=0, res1=None, res2=None, up_log2=0,
                out_nchw=False, res1_shift=0, res1_off_w=0):
    """The same op through the C ABI (pp_conv2d, host buffers)."""
    cout, cin, kh, kw = weight.shape
    n, h, w, cx = x.shape
    assert cx == cin and cin % 4 == 0
    W, b = pack_conv(weight, bias, cin_pad=cin)
    op = L.pp_op(type=L.PP_OP_CONV, in_=0, out=0, res1=-1, res2=-1, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride,
                 pad_h=pad[0], pad_w=pad[1], dil_h=dil[0], dil_w=dil[1], relu=relu, up_log2=up_log2,
                 out_nchw=int(out_nchw), res1_shift=res1_shift, res1_off_w=res1_off_w, w_off=0, b_off=0)
    ho = clib.out_dim(h, kh, stride, pad[0], dil[0]) << up_log2
    wo = clib.out_dim(w, kw, stride, pad[1], dil[1]) << up_log2
    y = np.full((n, cout, ho, wo) if out_nchw else (n, ho, wo, cout), np.nan, dtype=np.float32)
    x = np.ascontiguousarray(x, np.float32)
    r1 = None if res1 is None else np.ascontiguousarray(res1, np.float32)
    r2 = None if res2 is None else np.ascontiguousarray(res2, np.float32)
    L.check(ctx.lib.pp_conv2d(ctx.handle, C.byref(op), n, h, w, L.ptr(x), L.ptr(W), L.ptr(b), L.ptr(r1), L.ptr(r2),
                              L.ptr(y), 0 if r1 is None else r1.shape[1], 0 if r1 is None else r1.shape[2],
                              L.PP_MEM_HOST), "pp_conv2d")
    return y
