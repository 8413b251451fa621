"""CPU: the C-ABI library loads and exports every symbol declared in include/vpu_hip.h (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vpu_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vpu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from pvpuformer_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vpu_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.load().vpu_abi_version() == 2


def test_gemm_desc_layout_matches_header():
    from pvpuformer_amd._lib import GemmDesc
    # 7 pointers, 10 int32, 8 int64, 5 int32, 3 floats, pointer + int64 + pointer, then (ABI 2) cs_tn / cs_t0 (2 int32) and
    # cs_ld (int64) -- must equal the C struct size
    assert ctypes.sizeof(GemmDesc) == 7 * 8 + 10 * 4 + 8 * 8 + 5 * 4 + 3 * 4 + 24 + 2 * 4 + 8
