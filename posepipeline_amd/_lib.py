"""ctypes binding of libposepipe_hip.so (include/posepipe_hip.h).

The HIP library is the product path; there is no CPU fallback.  Importing this module works
without a GPU (so that the ABI can be inspected on a build box), but every compute entry point
needs a context, and ``Context()`` raises when no MI355X is visible or the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POSEPIPE_LIB", os.path.join(_HERE, "libposepipe_hip.so"))   # override: kernel A/B builds

PP_MEM_HOST, PP_MEM_DEVICE = 0, 1
PP_OP_CONV, PP_OP_MAXPOOL, PP_OP_ROIALIGN, PP_OP_COPY, PP_OP_VIT_ENCODER, PP_OP_DEPTH_TO_SPACE, PP_OP_UPSAMPLE_ADD = 1, 2, 3, 4, 5, 6, 7
PP_OP_DECONV_BF16 = 8
PP_OP_AVGPOOL = 9
PP_RELU_NONE, PP_RELU_LAST, PP_RELU_FIRST = 0, 1, 2
PP_ACT_LEAKY, PP_ACT_MISH, PP_ACT_ELU, PP_ACT_SWISH = 3, 4, 5, 6
PP_NET_NUMERICS_DEFAULT, PP_NET_NUMERICS_EXACT, PP_NET_NUMERICS_SPLIT = 0, 1, 2
PP_NET_NUMERICS_SPLIT_BF16, PP_NET_NUMERICS_SPLIT_F16 = 3, 4
# "split": the form pp_conv_split_kind / POSEPIPE_SPLIT_F16 select when the net is created; the two forms by name
NUMERICS = {None: 0, "default": 0, "exact": 1, "split": 2, "split_bf16": 3, "split_f16": 4}


class PosePipeHipError(RuntimeError):
    pass


class pp_op(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("in_", C.c_int32),
        ("out", C.c_int32),
        ("res1", C.c_int32),
        ("res2", C.c_int32),
        ("cin", C.c_int32),
        ("cout", C.c_int32),
        ("kh", C.c_int32),
        ("kw", C.c_int32),
        ("stride", C.c_int32),
        ("pad_h", C.c_int32),
        ("pad_w", C.c_int32),
        ("dil_h", C.c_int32),
        ("dil_w", C.c_int32),
        ("relu", C.c_int32),
        ("up_log2", C.c_int32),
        ("out_nchw", C.c_int32),
        ("res1_shift", C.c_int32),
        ("res1_off_w", C.c_int32),
        ("out_c_off", C.c_int32),
        ("in_c_off", C.c_int32),
        ("pad_end", C.c_int32),
        ("w_off", C.c_int64),
        ("b_off", C.c_int64),
        ("in2", C.c_int32),          # ABI 7: PP_OP_UPSAMPLE_ADD's further coarse inputs (-1 = none)
        ("in3", C.c_int32),
        ("up2_log2", C.c_int32),
        ("up3_log2", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        kw.setdefault("in2", -1)
        kw.setdefault("in3", -1)
        super().__init__(*args, **kw)


class pp_buf(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("pad", C.c_int32)]   # pad: zero halo (ABI v5)


_vp = C.c_void_p
_i = C.c_int
_sz = C.c_size_t

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "pp_abi_version": (_i, []),
    "pp_last_error": (C.c_char_p, []),
    "pp_device_count": (_i, []),
    "pp_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "pp_ctx_destroy": (None, [_vp]),
    "pp_ctx_set_stream": (_i, [_vp, _vp]),
    "pp_ctx_synchronize": (_i, [_vp]),
    "pp_timer_start": (_i, [_vp]),
    "pp_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "pp_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "pp_free": (_i, [_vp, _vp]),
    "pp_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "pp_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "pp_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "pp_host_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "pp_host_free": (_i, [_vp, _vp]),
    "pp_upload_begin": (_i, [_vp, _vp, _vp, _sz]),
    "pp_upload_begin_nv12": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "pp_nv12_to_bgr": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pp_upload_wait": (_i, [_vp, _i]),
    "pp_upload_release": (_i, [_vp]),
    "pp_net_create": (_i, [_vp, C.POINTER(pp_op), _i, C.POINTER(pp_buf), _i, _vp, _sz, _i, C.POINTER(_vp)]),
    "pp_net_create_mem": (_i, [_vp, C.POINTER(pp_op), _i, C.POINTER(pp_buf), _i, _vp, _sz, _i, _i, C.POINTER(_vp)]),
    "pp_net_create_ex": (_i, [_vp, C.POINTER(pp_op), _i, C.POINTER(pp_buf), _i, _vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "pp_net_numerics": (_i, [_vp]),
    "pp_net_split_kind": (_i, [_vp]),
    "pp_conv_split_kind": (_i, [_i]),
    "pp_net_destroy": (None, [_vp]),
    "pp_net_buffer": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "pp_net_run": (_i, [_vp, _i, _i, _i]),
    "pp_net_forward": (_i, [_vp, _i, _i, _vp, _i, _vp, _i]),
    "pp_net_capture": (_i, [_vp, _i]),
    "pp_net_set_lanes": (_i, [_vp, _i]),
    "pp_net_set_lane_count": (_i, [_vp, _i]),
    "pp_net_conv_kinds": (_i, [_vp, _vp]),
    "pp_net_profile": (_i, [_vp, _i, _vp]),
    "pp_crop_resize_bilinear": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "pp_conv_force": (_i, [_i, _i]),
    "pp_conv_variant": (_i, [_i]),
    "pp_conv_exact": (_i, [_i]),
    "pp_conv2d": (_i, [_vp, C.POINTER(pp_op), _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i]),
    "pp_crop_affine_normalize": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    "pp_topdown_create": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, C.POINTER(_vp)]),
    "pp_topdown_destroy": (None, [_vp]),
    "pp_topdown_run": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "pp_topdown_run_precropped": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i]),
    "pp_topdown_timing": (_i, [_vp, _vp]),
    "pp_detector_input_size": (_i, [_i, _i] + [C.POINTER(C.c_int32)] * 4),
    "pp_detector_constants": (_i, [C.POINTER(C.c_double), _i]),
    "pp_debug_knob": (_i, [C.c_char_p, _i]),
    "pp_net_input_amax": (_i, [_vp, _i, C.POINTER(_vp)]),
    "pp_rescale_size": (_i, [_i, _i, _i, _i, _i] + [C.POINTER(C.c_int32)] * 4),
    "pp_resize_pad_normalize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, C.c_float, _vp]),
    "pp_detector_create": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(_vp)]),
    "pp_detector_destroy": (None, [_vp]),
    "pp_detector_run": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "pp_detector_timing": (_i, [_vp, _vp]),
    "pp_detector_enqueue": (_i, [_vp, _vp, _i, _i, _i]),
    "pp_detector_collect": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "pp_detector_enable_margins": (_i, [_vp, _i, C.c_float]),
    "pp_detector_margins": (_i, [_vp, _i, _vp]),
    "pp_nms": (_i, [_vp, _vp, _vp, _i, C.c_double, _i, _vp, C.POINTER(C.c_int32), _i]),
    "pp_videopose3d_lift": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "pp_letterbox_bicubic": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "pp_yolo_decode": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i]),
    "pp_reid_patches": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "pp_tracker_create": (_i, [_i, _i, C.c_double, C.c_double, _i, _i, C.POINTER(_vp)]),
    "pp_tracker_destroy": (None, [_vp]),
    "pp_tracker_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "pp_tracker_dump": (_i, [_vp, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "pp_kalman_initiate": (_i, [_vp, _vp, _vp]),
    "pp_kalman_predict": (_i, [_vp, _vp]),
    "pp_kalman_update": (_i, [_vp, _vp, _vp]),
    "pp_kalman_gating_distance": (_i, [_vp, _vp, _vp, _i, _vp]),
    "pp_linear_sum_assignment": (_i, [_vp, _i, _i, _vp, _vp, C.POINTER(C.c_int32)]),
    "pp_flip_merge_decode": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i]),
    "pp_net_vit_timing": (_i, [_vp, _i, _vp, _vp]),
    "pp_f32_to_bf16": (_i, [_vp, _vp, _vp, _sz]),
    "pp_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i]),
    "pp_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_float, _vp, _i]),
    "pp_attention_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


def load_library() -> C.CDLL:
    """Load libposepipe_hip.so; raises PosePipeHipError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PosePipeHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C posepipeline_amd/csrc` (there is no CPU fallback)"
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


import contextlib


@contextlib.contextmanager
def default_numerics(mode):
    """Process-wide default numerics ("exact" / "split" / None = the environment's) for the nets CREATED inside the block -- composite
    objects (Cascade, Detector, TopDown wrappers) create theirs in their constructors.  A net keeps what it was created with.
    NOT thread-safe (one process-wide switch): code that builds nets on several threads passes `numerics=` to the constructors
    (Net, Cascade, Detector, the YOLO / ReID encoders) instead; this context manager is for single-threaded tests."""
    lib = load_library()
    prev = _DEFAULT_NUMERICS[0]

    def apply(m):
        # "split_bf16" / "split_f16" also pin the split FORM (pp_conv_split_kind); the other modes leave it to the environment
        check(lib.pp_conv_exact({None: -1, "default": -1, "exact": 1, "split": 0, "split_bf16": 0, "split_f16": 0}[m]), "pp_conv_exact")
        check(lib.pp_conv_split_kind({"split_bf16": 0, "split_f16": 1}.get(m, -1)), "pp_conv_split_kind")
    apply(mode)
    _DEFAULT_NUMERICS[0] = mode
    try:
        yield
    finally:
        apply(prev)
        _DEFAULT_NUMERICS[0] = prev


_DEFAULT_NUMERICS = [None]


def last_error() -> str:
    return load_library().pp_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise PosePipeHipError(f"{what or 'libposepipe_hip'} failed (status {rc}): {last_error()}")


def ptr(a):
    """void* of a numpy array / int address / None / torch tensor."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous(), "tensor must be contiguous"
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def mem_kind(a) -> int:
    if isinstance(a, np.ndarray):
        return PP_MEM_HOST
    if hasattr(a, "is_cuda"):
        return PP_MEM_DEVICE if a.is_cuda else PP_MEM_HOST
    raise TypeError(type(a))


class Context:
    """One device + stream (pp_ctx).  Raises when no GPU is visible: the product path never
    degrades to a CPU implementation."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        check(self.lib.pp_ctx_create(device, C.byref(h)), "pp_ctx_create")
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.pp_ctx_synchronize(self.handle), "pp_ctx_synchronize")

    def timer_start(self):
        check(self.lib.pp_timer_start(self.handle), "pp_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.lib.pp_timer_stop(self.handle, C.byref(ms)), "pp_timer_stop")
        return float(ms.value)

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(self.lib.pp_malloc(self.handle, nbytes, C.byref(p)), "pp_malloc")
        return int(p.value)

    def free(self, dptr: int):
        check(self.lib.pp_free(self.handle, C.c_void_p(dptr)), "pp_free")

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(self.lib.pp_memcpy_h2d(self.handle, C.c_void_p(dptr), ptr(arr), arr.nbytes), "pp_memcpy_h2d")

    def d2d(self, dst: int, src: int, nbytes: int):
        """asynchronous device-to-device copy on the context's stream"""
        check(self.lib.pp_memcpy_d2d(self.handle, C.c_void_p(dst), C.c_void_p(src), nbytes), "pp_memcpy_d2d")

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags["C_CONTIGUOUS"]
        check(self.lib.pp_memcpy_d2h(self.handle, ptr(arr), C.c_void_p(dptr), arr.nbytes), "pp_memcpy_d2h")
