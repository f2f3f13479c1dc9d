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
    assert _lib.load_library().pp_abi_version() == 10


def test_struct_layout_matches_header():
    # pp_op (ABI v7): 22 int32 + 2 int64 + 4 int32 (in2, in3, up2_log2, up3_log2)
    assert ctypes.sizeof(_lib.pp_op) == 120
    assert _lib.pp_op.w_off.offset == 88 and _lib.pp_op.b_off.offset == 96 and _lib.pp_op.in2.offset == 104
    assert _lib.pp_op().in2 == -1 and _lib.pp_op().in3 == -1            # "none" by default, like res1 / res2 in every builder
    assert ctypes.sizeof(_lib.pp_buf) == 16            # h, w, c, pad (ABI v5)


def test_no_gpu_means_loud_failure():
    lib = _lib.load_library()
    if lib.pp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.PosePipeHipError, match="no HIP device|no CPU fallback"):
        _lib.Context(0)


def test_missing_library_means_loud_failure(tmp_path):
    """the product path has no fallback: without libposepipe_hip.so the wrappers raise instead of computing on the CPU"""
    import os
    import subprocess
    import sys
    code = ("import os\n"
            "from posepipeline_amd import _lib\n"
            "from posepipeline_amd.wrappers import mmpose\n"
            "try:\n"
            "    mmpose._model('HRNet_W32_COCO')\n"
            "except _lib.PosePipeHipError as e:\n"
            "    print('LOUD:', e)\n")
    env = dict(os.environ, POSEPIPE_LIB=str(tmp_path / "absent.so"), POSEPIPE_SYNTHETIC_WEIGHTS="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert "LOUD:" in out.stdout and "absent.so" in out.stdout, out.stdout + out.stderr
