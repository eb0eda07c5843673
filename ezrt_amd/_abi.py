"""ctypes declarations for the two C ABIs (include/ezrt.h, include/ezrt_scene_c.h).

`declare_trace_abi(lib)` attaches argtypes/restypes for every symbol of
include/ezrt.h to an already-opened CDLL.  The product only ever opens
ezrt_amd/lib/libezrt_hip.so (see `load_hip`); tests open the CPU oracle
themselves and reuse `declare_trace_abi` so both sides share one binding.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_uint64_p = C.POINTER(C.c_uint64)
c_int64_p = C.POINTER(C.c_int64)

EZRT_CTR_COUNT = 8
CTR_NAMES = ("rays", "node_pops", "inner_pops", "tri_tests", "mat_fetch", "samples", "env_map", "env_cache")

INTEGRATOR_P3_DIFFUSE = 3
INTEGRATOR_P4_DISNEY = 4
INTEGRATOR_P5_SOBOL = 50
INTEGRATOR_P5_MIS = 51
INTEGRATOR_P5_MIS_ANISO = 52  # P5's loop with the anisotropic lobe evaluated + importance-sampled
FILTER_NEAREST = 0
FILTER_BILINEAR = 1


class EzrtRenderParams(C.Structure):
    """Mirror of `EzrtRenderParams` (include/ezrt.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
        ("frame0", C.c_uint32), ("spp", C.c_uint32),
        ("max_bounce", C.c_int32), ("integrator", C.c_int32),
        ("eye", C.c_float * 3),
        ("camera_rotate", C.c_float * 16),
        ("env_clamp", C.c_float),
        ("tile_w", C.c_int32), ("tile_h", C.c_int32),
        ("shard_index", C.c_int32), ("shard_count", C.c_int32),
    ]


# every symbol include/ezrt.h declares: name -> (restype, argtypes)
TRACE_ABI = {
    "ezrt_scene_create": (C.c_int, [c_float_p, C.c_int, c_float_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ezrt_scene_destroy": (None, [C.c_void_p]),
    "ezrt_scene_set_env": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int]),
    "ezrt_scene_set_sampler": (C.c_int, [C.c_void_p, C.c_int]),
    "ezrt_render": (C.c_int, [C.c_void_p, C.POINTER(EzrtRenderParams), c_float_p]),
    "ezrt_render_device": (C.c_int, [C.c_void_p, C.POINTER(EzrtRenderParams), C.c_void_p, C.c_void_p]),
    "ezrt_frame_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ezrt_frame_destroy": (C.c_int, [C.c_void_p]),
    "ezrt_frame_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p]),
    "ezrt_frame_write": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p]),
    "ezrt_render_paths": (C.c_int, [C.c_void_p, C.POINTER(EzrtRenderParams), c_int32_p, c_float_p, c_float_p]),
    "ezrt_frame_nonfinite": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, c_int64_p]),
    "ezrt_query_hits": (C.c_int, [C.c_void_p, c_float_p, C.c_int, c_int32_p, c_float_p]),
    "ezrt_tonemap": (C.c_int, [c_float_p, C.c_int, c_uint8_p]),
    "ezrt_sobol": (C.c_int, [C.c_uint32, C.c_int, C.c_int, c_float_p]),
    "ezrt_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ezrt_set_instrumentation": (C.c_int, [C.c_void_p, C.c_int]),
    "ezrt_counters": (C.c_int, [C.c_void_p, c_uint64_p]),
    "ezrt_counters_reset": (C.c_int, [C.c_void_p]),
    "ezrt_last_render_ms": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.POINTER(C.c_int)]),
    "ezrt_scene_stats": (C.c_int, [C.c_void_p, c_int64_p]),
    "ezrt_scene_prune_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "ezrt_debug_math": (C.c_int, [C.c_int, c_float_p, c_float_p, C.c_int, c_float_p]),
    "ezrt_last_error": (C.c_char_p, []),
    "ezrt_trim": (C.c_int, []),
    "ezrt_backend": (C.c_char_p, []),
}

HOST_ABI = {
    "ezrt_host_scene_new": (C.c_void_p, []),
    "ezrt_host_scene_free": (None, [C.c_void_p]),
    "ezrt_host_material_defaults": (C.c_int, [C.c_int, c_float_p]),
    "ezrt_host_get_transform_matrix": (C.c_int, [c_float_p, c_float_p, c_float_p, c_float_p]),
    "ezrt_host_read_obj": (C.c_int, [C.c_void_p, C.c_char_p, c_float_p, c_float_p, C.c_int]),
    "ezrt_host_read_obj_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, c_float_p, c_float_p, C.c_int]),
    "ezrt_host_add_triangles": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "ezrt_host_build_bvh": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ezrt_host_set_tie_order": (C.c_int, [C.c_int]),
    "ezrt_host_build_stats": (C.c_int, [C.c_void_p, c_int64_p]),
    "ezrt_host_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ezrt_host_encode": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "ezrt_host_hdr_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(c_float_p)]),
    "ezrt_host_hdr_load_memory": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(c_float_p)]),
    "ezrt_host_hdr_cache": (C.c_int, [c_float_p, C.c_int, C.c_int, c_float_p]),
    "ezrt_host_free": (None, [C.c_void_p]),
    "ezrt_host_camera": (C.c_int, [C.c_float, C.c_float, C.c_float, c_float_p, c_float_p]),
    "ezrt_host_p2_query": (C.c_int, [c_float_p, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, C.c_int, c_float_p,
                                     C.POINTER(C.c_int), c_float_p]),
    "ezrt_host_last_error": (C.c_char_p, []),
}


# GPU scene-build entry points of libezrt_hip.so only (include/ezrt_build.h)
BUILD_ABI = {
    "ezrt_build_lbvh": (C.c_int, [c_float_p, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, C.POINTER(C.c_int),
                                  c_float_p]),
    "ezrt_build_sah": (C.c_int, [c_float_p, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, C.POINTER(C.c_int),
                                 c_float_p]),
    "ezrt_build_median": (C.c_int, [c_float_p, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, C.POINTER(C.c_int),
                                    c_float_p]),
    "ezrt_build_device_count": (C.c_int, []),
}


# include/ezrt_mgpu.h: one process, N devices (both libraries: the oracle's "devices" are host memory)
MGPU_ABI = {
    "ezrt_mgpu_create": (C.c_int, [c_float_p, C.c_int, c_float_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                   C.POINTER(C.c_void_p)]),
    "ezrt_mgpu_destroy": (None, [C.c_void_p]),
    "ezrt_mgpu_set_env": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int]),
    "ezrt_mgpu_set_sampler": (C.c_int, [C.c_void_p, C.c_int]),
    "ezrt_mgpu_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ezrt_mgpu_render": (C.c_int, [C.c_void_p, C.POINTER(EzrtRenderParams)]),
    "ezrt_mgpu_gather": (C.c_int, [C.c_void_p, c_float_p]),
    "ezrt_mgpu_frame_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ezrt_mgpu_counters": (C.c_int, [C.c_void_p, c_uint64_p]),
    "ezrt_mgpu_last_ms": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_int64_p]),
    "ezrt_tiles_packed_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ezrt_tiles_pack_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "ezrt_tiles_unpack_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p]),
}
TRANSPORT_RCCL, TRANSPORT_PEER, TRANSPORT_HOST = 0, 1, 2


def _declare(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def declare_trace_abi(lib):
    return _declare(_declare(lib, TRACE_ABI), MGPU_ABI)


def declare_host_abi(lib):
    return _declare(lib, HOST_ABI)


_hip = None
_host = None


def load_hip():
    """Open the HIP product library.  There is no CPU fallback: if the extension
    is missing this raises."""
    global _hip
    if _hip is None:
        path = os.environ.get("EZRT_HIP_LIB") or os.path.join(LIB_DIR, "libezrt_hip.so")  # override: A/B builds
        if not os.path.exists(path):
            raise RuntimeError(
                "ezrt_amd: %s is missing -- build it with `make hip` (or __graft_entry__.build()); "
                "there is no CPU fallback for the trace" % path)
        _hip = _declare(declare_trace_abi(C.CDLL(path)), BUILD_ABI)
    return _hip


def load_host():
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libezrt_scene.so")
        if not os.path.exists(path):
            raise RuntimeError("ezrt_amd: %s is missing -- build it with `make host`" % path)
        _host = declare_host_abi(C.CDLL(path))
    return _host
