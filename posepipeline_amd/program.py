"""Layer programs: the host-side description that pp_net_create() compiles.

A backbone is a straight-line list of pp_op records over NHWC fp32 activation buffers plus one flat
weight blob.  ProgramBuilder is used by posepipeline_amd/models/*.py to turn an architecture spec +
a state_dict (torch layouts, mmpose / VideoPose3D key names) into that form:
  * BatchNorm is folded into (weight, bias) here -- float64 math, rounded once to float32;
  * conv weights are re-laid out to W[K/32][cout_pad16][32] (k = (kh*KW + kw)*cin_pad4 + cin, chunks of 32 k's in
    MFMA operand order, see pack_conv);
  * virtual buffers get physical ids by a linear-scan over lifetimes (exact-shape pooling).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L


def fold_bn(weight, conv_bias, gamma, beta, mean, var, eps=1e-5):
    """Fold eval-mode BatchNorm into the preceding conv.  float64, one rounding to float32."""
    scale = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps)
    w = weight.astype(np.float64) * scale.reshape((-1,) + (1,) * (weight.ndim - 1))
    b0 = np.zeros_like(scale) if conv_bias is None else conv_bias.astype(np.float64)
    b = beta.astype(np.float64) + (b0 - mean.astype(np.float64)) * scale
    return w.astype(np.float32), b.astype(np.float32)


def pack_conv(weight, bias, cin_pad=None):
    """[cout][cin][kh][kw] (or [cout][cin][k] for Conv1d) -> (W[Kpad32/32][cout_pad16][32], bias[cout_pad16])."""
    w = np.asarray(weight, dtype=np.float32)
    if w.ndim == 3:  # Conv1d: treat as kh = 1
        w = w[:, :, None, :]
    cout, cin, kh, kw = w.shape
    cin_p = cin_pad if cin_pad is not None else (cin + 3) // 4 * 4
    assert cin_p % 4 == 0 and cin_p >= cin
    cout_p = (cout + 15) // 16 * 16
    k = kh * kw * cin_p
    k_p = (k + 31) // 32 * 32
    wk = np.zeros((kh, kw, cin_p, cout_p), dtype=np.float32)
    wk[:, :, :cin, :cout] = np.transpose(w, (2, 3, 1, 0))
    flat = np.zeros((k_p, cout_p), dtype=np.float32)            # [k][cout], k = (kh*KW + kw)*cin_pad + cin
    flat[:k] = wk.reshape(k, cout_p)
    # chunks of 32 k's, per output channel in MFMA operand order: position 8*g + s holds k = 4*s + g
    chunks = flat.reshape(k_p // 32, 8, 4, cout_p)              # [chunk][s][g][cout]
    out = np.ascontiguousarray(np.transpose(chunks, (0, 3, 2, 1))).reshape(k_p // 32, cout_p, 32)
    b = np.zeros((cout_p,), dtype=np.float32)
    if bias is not None:
        b[:cout] = np.asarray(bias, dtype=np.float32)
    return out, b


@dataclass
class VBuf:
    h: int
    w: int
    c: int
    pinned: bool = False  # inputs / outputs keep their own physical buffer


@dataclass
class Program:
    ops: list
    bufs: list                      # physical (h, w, c)
    blob: np.ndarray
    named: dict = field(default_factory=dict)   # name -> physical buffer id
    buf_pad: list = field(default_factory=list)   # physical zero halo per buffer (pp_buf.pad), [] = all dense
    flops: float = 0.0              # algorithmic FLOPs per sample (2 * MACs of the real, unpadded convs)
    op_flops: list = field(default_factory=list)
    op_names: list = field(default_factory=list)


class ProgramBuilder:
    def __init__(self):
        self.vbufs: list[VBuf] = []
        self.vops: list[dict] = []
        self.blob_parts: list[np.ndarray] = []
        self.blob_len = 0
        self.named: dict[str, int] = {}

    # ---- buffers -----------------------------------------------------------------------------
    def buf(self, h, w, c, name=None, pinned=False) -> int:
        self.vbufs.append(VBuf(int(h), int(w), int(c), pinned or name is not None))
        vid = len(self.vbufs) - 1
        if name is not None:
            self.named[name] = vid
        return vid

    def dims(self, vid):
        b = self.vbufs[vid]
        return b.h, b.w, b.c

    def _add_blob(self, arr) -> int:
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self.blob_len
        pad = (-arr.size) % 4
        self.blob_parts.append(arr)
        if pad:
            self.blob_parts.append(np.zeros(pad, np.float32))
        self.blob_len += arr.size + pad
        return off

    # ---- ops ---------------------------------------------------------------------------------
    def conv(self, x, weight, bias, *, stride=1, pad=(0, 0), dil=(1, 1), relu=L.PP_RELU_NONE, res1=-1, res2=-1,
             up_log2=0, out=None, out_nchw=False, res1_shift=0, res1_off_w=0, out_c_off=0, pad_end=(0, 0),
             front_only=(0, 0), name="conv") -> int:
        """weight: torch layout, BN already folded.  Returns the (virtual) output buffer.
        out / out_c_off: write channels [out_c_off, out_c_off + cout) of an existing wider buffer (Concatenate).
        pad_end: (rows, cols) in {0, 1}: one extra zero row / column at the bottom / right (TensorFlow SAME).
        front_only: (rows, cols) in {0, 1}: `pad` applies in front only (the last output row / column is dropped)."""
        if isinstance(pad, int):
            pad = (pad, pad)
        if isinstance(dil, int):
            dil = (dil, dil)
        h, w, cin_buf = self.dims(x)
        wt = np.asarray(weight)
        if wt.ndim == 3:
            wt = wt[:, :, None, :]
        cout, cin, kh, kw = wt.shape
        assert cin <= cin_buf and cin_buf % 4 == 0, (cin, cin_buf)
        W, b = pack_conv(wt, bias, cin_pad=cin_buf)
        ho = (h + pad_end[0] + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride + 1 - front_only[0]
        wo = (w + pad_end[1] + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride + 1 - front_only[1]
        if out is None:
            assert out_c_off == 0
            out = self.buf(ho << up_log2, wo << up_log2, cout)
        else:
            oh, ow, oc = self.dims(out)
            assert (oh, ow) == (ho << up_log2, wo << up_log2) and out_c_off + cout <= oc and out_c_off % 4 == 0, \
                (self.dims(out), ho, wo, cout, out_c_off)
        w_off = self._add_blob(W)
        b_off = self._add_blob(b)
        self.vops.append(dict(type=L.PP_OP_CONV, in_=x, out=out, res1=res1, res2=res2, cin=cin_buf, cout=cout, kh=kh,
                              kw=kw, stride=stride, pad_h=pad[0], pad_w=pad[1], dil_h=dil[0], dil_w=dil[1], relu=relu,
                              up_log2=up_log2, out_nchw=int(out_nchw), res1_shift=res1_shift, res1_off_w=res1_off_w,
                              out_c_off=out_c_off, in_c_off=0,
                              pad_end=pad_end[0] | (pad_end[1] << 1) | (front_only[0] << 2) | (front_only[1] << 3),
                              w_off=w_off, b_off=b_off, name=name,
                              flops=2.0 * ho * wo * cout * cin * kh * kw))
        return out

    def maxpool(self, x, k, stride, pad, name="maxpool", out=None, out_c_off=0, in_c_off=0, c=None) -> int:
        """c / in_c_off: pool channels [in_c_off, in_c_off + c) of x; out / out_c_off: into a slice of a wider buffer."""
        h, w, cx = self.dims(x)
        c = cx if c is None else c
        assert in_c_off + c <= cx and c % 4 == 0 and in_c_off % 4 == 0
        ho = (h + 2 * pad - (k - 1) - 1) // stride + 1
        wo = (w + 2 * pad - (k - 1) - 1) // stride + 1
        if out is None:
            assert out_c_off == 0
            out = self.buf(ho, wo, c)
        else:
            oh, ow, oc = self.dims(out)
            assert (oh, ow) == (ho, wo) and out_c_off + c <= oc and out_c_off % 4 == 0
        self.vops.append(dict(type=L.PP_OP_MAXPOOL, in_=x, out=out, res1=-1, res2=-1, cin=c, cout=c, kh=k, kw=k,
                              stride=stride, pad_h=pad, pad_w=pad, dil_h=1, dil_w=1, relu=0, up_log2=0, out_nchw=0,
                              res1_shift=0, res1_off_w=0, out_c_off=out_c_off, in_c_off=in_c_off, pad_end=0,
                              w_off=0, b_off=0, name=name, flops=0.0))
        return out

    def avgpool(self, x, kh, kw, stride=1, name="avgpool") -> int:
        """nn.AvgPool2d((kh, kw), stride) without padding"""
        h, w, c = self.dims(x)
        out = self.buf((h - kh) // stride + 1, (w - kw) // stride + 1, c)
        return self._plain_op(L.PP_OP_AVGPOOL, x, out, cin=c, cout=c, kh=kh, kw=kw, stride=stride, name=name)

    def _plain_op(self, type_, x, out, *, cin, cout, kh=1, kw=1, stride=1, w_off=0, name="op", flops=0.0):
        self.vops.append(dict(type=type_, in_=x, out=out, res1=-1, res2=-1, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride,
                              pad_h=0, pad_w=0, dil_h=1, dil_w=1, relu=0, up_log2=0, out_nchw=0, res1_shift=0,
                              res1_off_w=0, out_c_off=0, in_c_off=0, pad_end=0, w_off=w_off, b_off=0, name=name,
                              flops=flops))
        return out

    def vit_encoder(self, x, params, *, depth, heads, mlp_ratio, name="vit_encoder") -> int:
        """PP_OP_VIT_ENCODER on a [h][w][dim] fp32 token map.  params: flat fp32 block in the layout of
        include/posepipe_hip.h (pos, depth x block, final LayerNorm)."""
        h, w, dim = self.dims(x)
        t, hid = h * w, dim * mlp_ratio
        per_block = 2 * dim + 3 * dim * dim + 3 * dim + dim * dim + dim + 2 * dim + hid * dim + hid + dim * hid + dim
        params = np.asarray(params, dtype=np.float32).reshape(-1)
        assert params.size == t * dim + per_block * depth + 2 * dim, (params.size, t, dim, depth)
        out = self.buf(h, w, dim)
        macs = depth * (t * (4 * dim * dim + 2 * dim * hid) + 2 * t * t * dim)
        return self._plain_op(L.PP_OP_VIT_ENCODER, x, out, cin=dim, cout=dim, kh=depth, kw=heads, stride=mlp_ratio,
                              w_off=self._add_blob(params), name=name, flops=2.0 * macs)

    def upsample_add(self, t, *, up_log2, res1=-1, res2=-1, relu=L.PP_RELU_NONE, name="upsample_add", more=()) -> int:
        """out = act((((res1 + up(t, 2^up_log2)) + up(t2, 2^u2)) + up(t3, 2^u3)) + res2): the accumulate step(s) of an HRNet fuse
        layer in one pass.  more: up to two further coarse terms [(buffer, up_log2), ...], added in this order."""
        h, w, c = self.dims(t)
        out = self.buf(h << up_log2, w << up_log2, c)
        more = list(more)
        assert len(more) <= 2
        for tb, ub in more:
            assert self.dims(tb) == ((h << up_log2) >> ub, (w << up_log2) >> ub, c), (self.dims(tb), self.dims(out), ub)
        (in2, up2), (in3, up3) = (more + [(-1, 0), (-1, 0)])[:2]
        self.vops.append(dict(type=L.PP_OP_UPSAMPLE_ADD, in_=t, out=out, res1=res1, res2=res2, cin=c, cout=c, kh=1, kw=1,
                              stride=1, pad_h=0, pad_w=0, dil_h=1, dil_w=1, relu=relu, up_log2=up_log2, out_nchw=0,
                              res1_shift=0, res1_off_w=0, out_c_off=0, in_c_off=0, pad_end=0, w_off=0, b_off=0, name=name,
                              in2=in2, in3=in3, up2_log2=up2, up3_log2=up3, flops=0.0))
        return out

    def deconv4x4s2_bf16(self, x, weight, bias, *, relu=L.PP_RELU_NONE, name="deconv_bf16") -> int:
        """ConvTranspose2d(kernel 4, stride 2, padding 1) (+ folded BN, ReLU) as ONE bf16 GEMM over the 16 kernel taps +
        a 4-term gather (PP_OP_DECONV_BF16).  weight: torch ConvTranspose2d layout [cin][cout][4][4], BN already folded."""
        h, w, cin_buf = self.dims(x)
        wt = np.asarray(weight, dtype=np.float32)
        cin, cout = wt.shape[:2]
        assert wt.shape[2:] == (4, 4) and cin == cin_buf and cin % 64 == 0 and cout % 8 == 0, (wt.shape, cin_buf)
        blocks = np.empty((2, 2, 2, 2, cout, cin), np.float32)           # [a][b][r][s][cout][cin]
        for a in (0, 1):
            for b in (0, 1):
                for r in (0, 1):
                    for s in (0, 1):
                        blocks[a, b, r, s] = wt[:, :, 3 - a - 2 * r, 3 - b - 2 * s].T
        bb = np.zeros(cout, np.float32) if bias is None else np.asarray(bias, np.float32)
        out = self.buf(2 * h, 2 * w, cout)
        w_off = self._add_blob(blocks)
        b_off = self._add_blob(bb)
        self.vops.append(dict(type=L.PP_OP_DECONV_BF16, in_=x, out=out, res1=-1, res2=-1, cin=cin, cout=cout, kh=4, kw=4,
                              stride=2, pad_h=1, pad_w=1, dil_h=1, dil_w=1, relu=relu, up_log2=0, out_nchw=0, res1_shift=0,
                              res1_off_w=0, out_c_off=0, in_c_off=0, pad_end=0, w_off=w_off, b_off=b_off, name=name,
                              flops=2.0 * h * w * cin * cout * 16))
        return out

    def depth_to_space(self, x, name="depth_to_space") -> int:
        """[h][w][4c] (channel groups g = 2*dy + dx) -> [2h][2w][c]"""
        h, w, c4 = self.dims(x)
        assert c4 % 16 == 0
        out = self.buf(2 * h, 2 * w, c4 // 4)
        return self._plain_op(L.PP_OP_DEPTH_TO_SPACE, x, out, cin=c4, cout=c4 // 4, name=name)

    def deconv4x4s2(self, x, weight, bias, *, relu=L.PP_RELU_NONE, name="deconv") -> int:
        """ConvTranspose2d(kernel 4, stride 2, padding 1) (+ folded BN, ReLU) as four 2x2 convolutions, one per output
        parity, written to the channel groups of a [h][w][4*cout] buffer, then depth_to_space.
        weight: torch ConvTranspose2d layout [cin][cout][4][4].  out[2j + a] = sum_i in[i] w[2j + a + 1 - 2i]:
        a = 0 -> taps (in[j-1], in[j]) x (w[3], w[1]), one zero row in front; a = 1 -> (in[j], in[j+1]) x (w[2], w[0]),
        one zero row behind."""
        h, w, cin_buf = self.dims(x)
        wt = np.asarray(weight, dtype=np.float32)
        cin, cout = wt.shape[:2]
        assert wt.shape[2:] == (4, 4) and cout % 4 == 0
        wide = self.buf(h, w, 4 * cout)
        taps = {0: (3, 1), 1: (2, 0)}
        for a in (0, 1):
            for b in (0, 1):
                sub = wt[:, :, list(taps[a]), :][:, :, :, list(taps[b])]          # [cin][cout][2][2]
                self.conv(x, np.transpose(sub, (1, 0, 2, 3)), bias, pad=(1 - a, 1 - b), pad_end=(a, b),
                          front_only=(1 - a, 1 - b), relu=relu, out=wide, out_c_off=(2 * a + b) * cout,
                          name=f"{name}.p{a}{b}")
        return self.depth_to_space(wide, name=name + ".d2s")

    # ---- finalize ----------------------------------------------------------------------------
    def _halo_plan(self):
        """Zero halo (pp_buf.pad) per virtual buffer.  A buffer that only convolutions touch and that a PADDED convolution reads
        is stored with `pad` zero columns / rows at the right / bottom of every row / image: the reader's taps then never
        need a bounds test (the largest share of the K loop's vector instructions, profiles/r02_conv_probe.txt).  Named
        (pinned) buffers stay dense: other kernels and the host see them.  POSEPIPE_CONV_HALO=0 turns the layout off."""
        import os
        n_v = len(self.vbufs)
        pad = [0] * n_v
        if os.environ.get("POSEPIPE_CONV_HALO", "1") == "0":
            return pad
        ok = [not b.pinned and b.c % 4 == 0 for b in self.vbufs]
        need = [0] * n_v
        for op in self.vops:
            conv = op["type"] == L.PP_OP_CONV
            for k in ("in_", "out", "res1", "res2", "in2", "in3"):
                v = op.get(k, -1)
                if v >= 0 and not conv:
                    ok[v] = False
            if not conv:
                continue
            o = op["out"]
            if op["out_nchw"] or op["out_c_off"] or self.vbufs[o].c != op["cout"]:
                ok[o] = False                       # channel slices / NCHW planes: dense
            h, w, _ = self.dims(op["in_"])
            eh, ew = op["pad_end"] & 1, (op["pad_end"] >> 1) & 1
            ho = (h + eh + 2 * op["pad_h"] - op["dil_h"] * (op["kh"] - 1) - 1) // op["stride"] + 1 - ((op["pad_end"] >> 2) & 1)
            wo = (w + ew + 2 * op["pad_w"] - op["dil_w"] * (op["kw"] - 1) - 1) // op["stride"] + 1 - ((op["pad_end"] >> 3) & 1)
            over = max(op["pad_h"], op["pad_w"], (ho - 1) * op["stride"] - op["pad_h"] + (op["kh"] - 1) * op["dil_h"] - (h - 1),
                       (wo - 1) * op["stride"] - op["pad_w"] + (op["kw"] - 1) * op["dil_w"] - (w - 1))
            if over > 0:
                need[op["in_"]] = max(need[op["in_"]], over)
        return [need[v] if ok[v] and 0 < need[v] <= 2 else 0 for v in range(n_v)]

    def build(self) -> Program:
        n_v = len(self.vbufs)
        last_use = [-1] * n_v
        first_def = [None] * n_v
        for i, op in enumerate(self.vops):
            for key in ("in_", "res1", "res2", "in2", "in3", "out"):
                v = op.get(key, -1)
                if v >= 0:
                    last_use[v] = i
            if first_def[op["out"]] is None:
                first_def[op["out"]] = i
        vpad = self._halo_plan()
        phys_dims: list[tuple] = []
        phys_pad: list[int] = []
        free: dict[tuple, list[int]] = {}
        v2p = [-1] * n_v
        # pinned buffers (inputs, named outputs) first, never recycled
        for v, b in enumerate(self.vbufs):
            if b.pinned:
                phys_dims.append((b.h, b.w, b.c))
                phys_pad.append(0)
                v2p[v] = len(phys_dims) - 1
        for i, op in enumerate(self.vops):
            v = op["out"]
            if v2p[v] < 0:
                key = self.dims(v) + (vpad[v],)
                pool = free.get(key, [])
                # never alias an operand of this very op
                busy = {v2p[op[k]] for k in ("in_", "res1", "res2", "in2", "in3") if op.get(k, -1) >= 0}
                pick = next((p for p in pool if p not in busy), None)
                if pick is not None:
                    pool.remove(pick)
                    v2p[v] = pick
                else:
                    phys_dims.append(key[:3])
                    phys_pad.append(vpad[v])
                    v2p[v] = len(phys_dims) - 1
            # release operands whose last use is this op
            for key in ("in_", "res1", "res2", "in2", "in3", "out"):
                u = op.get(key, -1)
                if u >= 0 and last_use[u] == i and not self.vbufs[u].pinned and v2p[u] >= 0:
                    lst = free.setdefault(self.dims(u) + (vpad[u],), [])
                    if v2p[u] not in lst:
                        lst.append(v2p[u])
        ops = []
        for op in self.vops:
            rec = L.pp_op()
            for k, val in op.items():
                if k in ("name", "flops"):
                    continue
                if k in ("in_", "out", "res1", "res2", "in2", "in3"):
                    val = v2p[val] if val >= 0 else -1
                setattr(rec, k, int(val))
            ops.append(rec)
        blob = np.concatenate(self.blob_parts) if self.blob_parts else np.zeros(4, np.float32)
        named = {k: v2p[v] for k, v in self.named.items()}
        return Program(ops=ops, bufs=phys_dims, buf_pad=phys_pad, blob=blob, named=named,
                       flops=float(sum(op["flops"] for op in self.vops)),
                       op_flops=[op["flops"] for op in self.vops], op_names=[op["name"] for op in self.vops])


class Net:
    """pp_net handle: resident weights + activation arena on one Context."""

    def __init__(self, ctx: L.Context, prog: Program, max_batch: int, blob_dev=None, numerics=None):
        """blob_dev: optional (device pointer, n_floats) of a weight blob that is already resident on ctx's device -- the
        tensor an RCCL broadcast delivered; prog.blob (host) is not read then.
        numerics: None / "default" (the process-wide pp_conv_exact / POSEPIPE_CONV_EXACT setting at this moment), "exact"
        (float32 MFMA kernels, the oracle's bits) or "split" (bf16 / fp16 matrix cores where eligible; "split_bf16" / "split_f16" name
        the form, "split" takes the process default, pp_conv_split_kind / POSEPIPE_SPLIT_F16).  Fixed for the net's life."""
        self.ctx = ctx
        self.prog = prog
        self.max_batch = int(max_batch)
        lib = ctx.lib
        n_ops = len(prog.ops)
        ops = (L.pp_op * n_ops)(*prog.ops)
        pads = prog.buf_pad or [0] * len(prog.bufs)
        bufs = (L.pp_buf * len(prog.bufs))(*[L.pp_buf(*d, p) for d, p in zip(prog.bufs, pads)])
        h = C.c_void_p()
        if blob_dev is not None:
            dptr, n_floats = blob_dev
            L.check(lib.pp_net_create_ex(ctx.handle, ops, n_ops, bufs, len(prog.bufs), C.c_void_p(int(dptr)), int(n_floats),
                                         L.PP_MEM_DEVICE, self.max_batch, L.NUMERICS[numerics], C.byref(h)), "pp_net_create_ex")
        else:
            blob = np.ascontiguousarray(prog.blob, dtype=np.float32)
            L.check(lib.pp_net_create_ex(ctx.handle, ops, n_ops, bufs, len(prog.bufs), L.ptr(blob), blob.size, L.PP_MEM_HOST,
                                         self.max_batch, L.NUMERICS[numerics], C.byref(h)), "pp_net_create_ex")
        self.handle = h
        self.numerics = "split" if lib.pp_net_numerics(h) == L.PP_NET_NUMERICS_SPLIT else "exact"
        # which split form the net runs ("split_bf16": three bf16 terms / six products, "split_f16": two fp16 terms / three products)
        self.split_kind = {L.PP_NET_NUMERICS_SPLIT_BF16: "split_bf16", L.PP_NET_NUMERICS_SPLIT_F16: "split_f16"}.get(lib.pp_net_split_kind(h), "exact")

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):      # a closed context already released the device
                self.ctx.lib.pp_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def buffer(self, name_or_id):
        """(device pointer, bytes per sample, (h, w, c))"""
        pid = self.prog.named[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        p = C.c_void_p()
        nb = C.c_size_t()
        L.check(self.ctx.lib.pp_net_buffer(self.handle, pid, C.byref(p), C.byref(nb)), "pp_net_buffer")
        return int(p.value), int(nb.value), self.prog.bufs[pid]

    def run(self, batch, first=0, last=None):
        last = len(self.prog.ops) if last is None else last
        L.check(self.ctx.lib.pp_net_run(self.handle, batch, first, last), "pp_net_run")

    def set_lanes(self, enable: bool):
        """multi-stream execution of independent ops on/off (off = serial launches, for additive kernel profiles)"""
        L.check(self.ctx.lib.pp_net_set_lanes(self.handle, int(bool(enable))), "pp_net_set_lanes")

    def set_lane_count(self, n_lanes: int):
        """how many HIP streams the program's independent ops are spread over (default 4; results identical for any count)"""
        L.check(self.ctx.lib.pp_net_set_lane_count(self.handle, int(n_lanes)), "pp_net_set_lane_count")

    def capture(self, batch):
        L.check(self.ctx.lib.pp_net_capture(self.handle, batch), "pp_net_capture")

    def forward(self, x: np.ndarray, in_name="input", out_name="output") -> np.ndarray:
        """Host convenience: x [n][h][w][c] fp32 NHWC -> named output buffer as numpy."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        pin, pout = self.prog.named[in_name], self.prog.named[out_name]
        assert tuple(x.shape[1:]) == tuple(self.prog.bufs[pin]), (x.shape, self.prog.bufs[pin])
        oh, ow, oc = self.prog.bufs[pout]
        out = np.empty((n, oh, ow, oc), dtype=np.float32)
        L.check(self.ctx.lib.pp_net_forward(self.handle, n, pin, L.ptr(x), pout, L.ptr(out), L.PP_MEM_HOST),
                "pp_net_forward")
        return out

    def read(self, name_or_id, batch) -> np.ndarray:
        dptr, nb, (h, w, c) = self.buffer(name_or_id)
        out = np.empty((batch, h, w, c), dtype=np.float32)
        self.ctx.d2h(out, dptr)
        return out

    def conv_kinds(self) -> np.ndarray:
        """per op: 0 not a conv, 1 float32 MFMA kernels, 2 bf16-split kernel (a property of the net, fixed at creation)"""
        kinds = np.zeros(len(self.prog.ops), dtype=np.int32)
        L.check(self.ctx.lib.pp_net_conv_kinds(self.handle, L.ptr(kinds)), "pp_net_conv_kinds")
        return kinds

    def profile(self, batch) -> np.ndarray:
        ms = np.zeros(len(self.prog.ops), dtype=np.float32)
        L.check(self.ctx.lib.pp_net_profile(self.handle, batch, L.ptr(ms)), "pp_net_profile")
        return ms
