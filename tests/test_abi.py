"""The C-ABI library loads without a GPU and exports every symbol include/posepipe_hip.h declares."""
import ctypes
import os
import re

import pytest

from posepipeline_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "posepipe_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_table_matches_header():
    names = set(declared_functions())
    bound = set(_lib.SIGNATURES)
    assert bound <= names, f"bound but not declared: {sorted(bound - names)}"
    # everything the Python host side binds must load
    _lib.load_library()
    assert _lib.load_library().pp_abi_version() == 3


def test_struct_layout_matches_header():
    # pp_op (ABI v2): 22 int32 + 2 int64
    assert ctypes.sizeof(_lib.pp_op) == 104
    assert _lib.pp_op.w_off.offset == 88 and _lib.pp_op.b_off.offset == 96
    assert ctypes.sizeof(_lib.pp_buf) == 12


def test_no_gpu_means_loud_failure():
    lib = _lib.load_library()
    if lib.pp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.PosePipeHipError, match="no HIP device|no CPU fallback"):
        _lib.Context(0)
