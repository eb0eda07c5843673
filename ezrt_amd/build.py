"""GPU scene build (include/ezrt_build.h): the linear BVH alternative to the host buildBVHwithSAH."""
import ctypes as C

import numpy as np

from . import _abi


def build_sah(tri36, leaf_n=8):
    """buildBVHwithSAH on the GPU: exactly the arrays of the host builder (HostScene.buildBVHwithSAH +
    encode), for scenes where the host build dominates.  -> (triangles [n, 36], nodes [m, 12], device ms)."""
    return _build("ezrt_build_sah", tri36, leaf_n)


def build_median(tri36, leaf_n=8):
    """buildBVH (median split) on the GPU: exactly the arrays of HostScene.buildBVH + encode."""
    return _build("ezrt_build_median", tri36, leaf_n)


def build_lbvh(tri36, leaf_n=8):
    """tri36 [n, 36] (reference triangle layout) -> (triangles reordered [n, 36], nodes [m, 12] in the
    reference node layout, device build time in ms).  Runs on the GPU; raises when the HIP library or
    the GPU is missing (no CPU fallback)."""
    return _build("ezrt_build_lbvh", tri36, leaf_n)


def device_count():
    """HIP devices visible to this process (0 on a CPU-only box)."""
    return int(_abi.load_hip().ezrt_build_device_count())


def _build(entry, tri36, leaf_n):
    lib = _abi.load_hip()
    tri = np.ascontiguousarray(tri36, np.float32).reshape(-1, 36)
    n = tri.shape[0]
    tri_out = np.zeros_like(tri)
    cap = max(2 * n, 2)
    nodes = np.zeros((cap, 12), np.float32)
    n_nodes = C.c_int(0)
    ms = C.c_float(0.0)
    fp = lambda a: a.ctypes.data_as(_abi.c_float_p)
    rc = getattr(lib, entry)(fp(tri), n, int(leaf_n), fp(tri_out), fp(nodes), cap, C.byref(n_nodes), C.byref(ms))
    if rc != 0:
        raise RuntimeError("%s: %s" % (entry, lib.ezrt_last_error().decode()))
    return tri_out, np.ascontiguousarray(nodes[:n_nodes.value]), float(ms.value)
