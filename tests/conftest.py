import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "ezrt_amd", "lib", "libezrt_scene.so"),
            os.path.join(ROOT, "ezrt_amd", "lib", "libezrt_hip.so"),
            os.path.join(ROOT, "oracle", "libezrt_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "host", "hip", "oracle", "examples"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle behind the same ctypes binding as the product (test infrastructure)."""
    _ensure_built()
    from ezrt_amd import _abi, trace
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libezrt_oracle.so"))
    return trace.TraceLib(_abi.declare_trace_abi(lib))


@pytest.fixture(scope="session")
def hip():
    """The product: libezrt_hip.so.  Fails loudly when there is no GPU."""
    _ensure_built()
    from ezrt_amd import trace
    return trace.hip()


@pytest.fixture(scope="session")
def bunny_small():
    _ensure_built()
    from ezrt_amd import scenes
    return scenes.bunny_scene(subdiv=0, want_cache=True)


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(REFERENCE)
