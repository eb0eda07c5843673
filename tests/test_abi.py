"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ezrt_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol(hip):
    names = _declared("ezrt.h")
    assert len(names) >= 17
    for n in names:
        assert hasattr(hip.lib, n), n
    assert hip.backend() == "hip:gfx950"


def test_oracle_library_exports_the_same_abi(oracle):
    for n in _declared("ezrt.h"):
        assert hasattr(oracle.lib, n), n
    assert oracle.backend() == "oracle:cpu"


def test_host_library_exports_every_declared_symbol():
    from ezrt_amd import _abi
    lib = _abi.load_host()
    names = _declared("ezrt_scene_c.h")
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_abi.HOST_ABI)


def test_binding_table_matches_header():
    from ezrt_amd import _abi
    assert set(_declared("ezrt.h")) == set(_abi.TRACE_ABI)


def test_mgpu_binding_table_matches_header_in_both_libraries(oracle):
    from ezrt_amd import _abi
    names = _declared("ezrt_mgpu.h")
    assert set(names) == set(_abi.MGPU_ABI) and len(names) == 13
    hip = _abi.load_hip()  # dlopen only
    for n in names:
        assert hasattr(hip, n) and hasattr(oracle.lib, n), n


def test_params_struct_layout_matches_header():
    from ezrt_amd import _abi
    # 6 ints + 2 uints + 2 ints + 3 + 16 floats + 1 float + 4 ints = 34 x 4 bytes
    assert ctypes.sizeof(_abi.EzrtRenderParams) == 34 * 4


def test_product_has_no_oracle_dependency():
    """Nothing under ezrt_amd/ may import, load or mention the oracle library."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "ezrt_amd")):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if "libezrt_oracle" in txt or "oracle/" in txt.replace("the oracle/", ""):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_hip_library_exports_the_gpu_build_entry_points(hip):
    from ezrt_amd import _abi
    names = _declared("ezrt_build.h")
    assert names and set(names) == set(_abi.BUILD_ABI)
    for n in names:
        assert hasattr(hip.lib, n), n
