"""numpy-facing wrapper over the trace C ABI (include/ezrt.h).

`TraceLib` wraps an opened + declared CDLL.  `hip()` returns the product
(libezrt_hip.so, gfx950 kernels).  The wrapper never picks a library itself and
has no fallback: a missing HIP library raises in `_abi.load_hip()`.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import EzrtRenderParams

_F = _abi.c_float_p


def _fp(a):
    return a.ctypes.data_as(_F)


def make_params(width, height, eye, camera_rotate, integrator, max_bounce, spp=1, frame0=0, rect=None,
                env_clamp=0.0, tile=(32, 32), shard=(0, 1)):
    p = EzrtRenderParams()
    p.width, p.height = int(width), int(height)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, width, height)
    p.x0, p.y0, p.x1, p.y1 = int(x0), int(y0), int(x1), int(y1)
    p.frame0, p.spp = int(frame0), int(spp)
    p.max_bounce, p.integrator = int(max_bounce), int(integrator)
    for i in range(3):
        p.eye[i] = float(eye[i])
    for i in range(16):
        p.camera_rotate[i] = float(camera_rotate[i])
    p.env_clamp = float(env_clamp)
    p.tile_w, p.tile_h = int(tile[0]), int(tile[1])
    p.shard_index, p.shard_count = int(shard[0]), int(shard[1])
    return p


class TraceError(RuntimeError):
    pass


class Scene:
    """An EzrtScene handle (device-resident scene replica)."""

    def __init__(self, tl, handle):
        self._tl = tl
        self._h = handle

    def close(self):
        if self._h:
            self._tl.lib.ezrt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise TraceError("%s (rc=%d)" % (self._tl.lib.ezrt_last_error().decode(), rc))

    def set_env(self, hdr, cache=None, filter=_abi.FILTER_BILINEAR):
        hdr = np.ascontiguousarray(hdr, np.float32)
        h, w, _ = hdr.shape
        cp = None
        if cache is not None:
            cache = np.ascontiguousarray(cache, np.float32)
            assert cache.shape == hdr.shape
            cp = _fp(cache)
        self._ck(self._tl.lib.ezrt_scene_set_env(self._h, _fp(hdr), cp, w, h, int(filter)))

    def set_sampler(self, sobol_dims):
        """8 (default) = the reference's Sobol table, dims wrap d & 7; 16 = eight more dimensions (include/ezrt.h)."""
        self._ck(self._tl.lib.ezrt_scene_set_sampler(self._h, int(sobol_dims)))

    def render(self, params, accum=None):
        """accum: float32 [H, W, 4] running mean (modified in place and returned)."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 4), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous
        assert accum.shape == (params.height, params.width, 4)
        self._ck(self._tl.lib.ezrt_render(self._h, C.byref(params), _fp(accum)))
        return accum

    def render_device(self, params, accum_ptr, stream=None):
        """accum_ptr: integer device address of an RGBA32F [H, W, 4] buffer; stream: hipStream_t int or None."""
        self._ck(self._tl.lib.ezrt_render_device(self._h, C.byref(params), C.c_void_p(accum_ptr),
                                                 C.c_void_p(stream or 0)))

    def render_paths(self, params, want_colour=True):
        slots = 1 + 2 * params.max_bounce
        tri = np.full((params.height, params.width, slots), -2, np.int32)
        t = np.zeros((params.height, params.width, slots), np.float32)
        col = np.zeros((params.height, params.width, 3), np.float32) if want_colour else None
        self._ck(self._tl.lib.ezrt_render_paths(self._h, C.byref(params), tri.ctypes.data_as(_abi.c_int32_p), _fp(t),
                                                _fp(col) if want_colour else None))
        return tri, t, col

    def query_hits(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self._ck(self._tl.lib.ezrt_query_hits(self._h, _fp(rays), n, tri.ctypes.data_as(_abi.c_int32_p), _fp(t)))
        return tri, t

    def set_option(self, name, value):
        self._ck(self._tl.lib.ezrt_set_option(self._h, name.encode(), int(value)))
        self._options = getattr(self, "_options", {})
        self._options[name] = int(value)   # (what THIS wrapper set: pipeline_selfcheck restores it)

    def set_instrumentation(self, level):
        self._ck(self._tl.lib.ezrt_set_instrumentation(self._h, int(level)))

    def counters(self):
        out = (C.c_uint64 * _abi.EZRT_CTR_COUNT)()
        self._ck(self._tl.lib.ezrt_counters(self._h, out))
        return dict(zip(_abi.CTR_NAMES, [int(x) for x in out]))

    def counters_reset(self):
        self._ck(self._tl.lib.ezrt_counters_reset(self._h))

    def last_render_ms(self):
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        self._ck(self._tl.lib.ezrt_last_render_ms(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def pipeline_selfcheck(self, params, accum_ptr, stream=None, calls=6, min_gain=1.01, bursts=3):
        """Is overlapping consecutive calls (option pipeline_calls) actually a gain HERE?  It leans on how the runtime maps streams onto
        hardware queues: in a host with many other streams (torch, RCCL) one of the library's chunk streams can land on the caller's
        queue, where the accumulation's barriers block the next chunk, and the overlap turns into a 1-3 % loss
        (profiles/r4/stream_pressure.txt).  This renders `calls` back-to-back calls of `params` into `accum_ptr`
        (a scratch frame buffer of the caller's: it is overwritten) twice with the knob on and twice with it off, interleaved, each
        burst closed by a synchronisation (ezrt_last_render_ms waits for the last call's end event), and KEEPS the knob on only if
        the pipelined bursts were at least `min_gain` times as fast (default 1.01: a tie is noise, and noise must not decide); otherwise
        it is switched off for this scene.  A knob the caller had switched OFF (set_option("pipeline_calls", 0) or EZRT_PIPELINE_CALLS=0)
        is respected: nothing is measured and it stays off; a value the caller had set (1 or 2) is the value restored when the check
        keeps the overlap.  Results never depend on the knob.  Returns {"ms_pipelined", "ms_plain", "gain", "kept"}.  A start-up
        step for a host that renders many frames."""
        import os
        import time
        prior = getattr(self, "_options", {}).get("pipeline_calls")
        if prior is None and os.environ.get("EZRT_PIPELINE_CALLS") is not None:
            prior = int(os.environ["EZRT_PIPELINE_CALLS"])
        if prior == 0:
            return {"ms_pipelined": None, "ms_plain": None, "gain": None, "kept": False, "skipped": "pipeline_calls was switched off by the caller"}

        def burst(on):
            self.set_option("pipeline_calls", 1 if on else 0)
            self.render_device(params, accum_ptr, stream)      # (scratch of the route in place, streams created)
            self.last_render_ms()
            t0 = time.perf_counter()
            for _ in range(int(calls)):
                self.render_device(params, accum_ptr, stream)
            self.last_render_ms()                               # blocks until the last call's end event
            return (time.perf_counter() - t0) * 1e3 / max(1, int(calls))

        on_ms, off_ms = [], []
        for _ in range(max(2, int(bursts))):
            on_ms.append(burst(True))
            off_ms.append(burst(False))
        ms_on, ms_off = min(on_ms), min(off_ms)
        gain = ms_off / ms_on if ms_on > 0 else 0.0
        kept = gain >= float(min_gain)
        self.set_option("pipeline_calls", (prior if prior else 1) if kept else 0)
        return {"ms_pipelined": ms_on, "ms_plain": ms_off, "gain": gain, "kept": kept}

    def prune_info(self):
        out = (C.c_double * 8)()
        self._ck(self._tl.lib.ezrt_scene_prune_info(self._h, out))
        return dict(zip(("mode", "G", "Z", "M", "unprunable_triangles", "margin_a", "retreed", "records4"), [float(x) for x in out]))

    def stats(self):
        out = (C.c_int64 * 6)()
        self._ck(self._tl.lib.ezrt_scene_stats(self._h, out))
        return dict(zip(("n_tri", "n_nodes", "depth", "n_leaves", "max_leaf", "device_bytes"), [int(x) for x in out]))


class Frame:
    """A device-resident lastFrame (ezrt_frame_*): RGBA32F [height, width, 4], zeroed."""

    def __init__(self, tl, width, height):
        self._tl = tl
        self.width, self.height = int(width), int(height)
        h = C.c_void_p()
        if tl.lib.ezrt_frame_create(self.width, self.height, C.byref(h)) != 0:
            raise TraceError(tl.lib.ezrt_last_error().decode())
        self.ptr = h.value

    def read(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        if self._tl.lib.ezrt_frame_read(self.ptr, self.width, self.height, _fp(out)) != 0:
            raise TraceError(self._tl.lib.ezrt_last_error().decode())
        return out

    def write(self, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        assert a.shape == (self.height, self.width, 4)
        if self._tl.lib.ezrt_frame_write(self.ptr, self.width, self.height, _fp(a)) != 0:
            raise TraceError(self._tl.lib.ezrt_last_error().decode())

    def close(self):
        if self.ptr:
            self._tl.lib.ezrt_frame_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TraceLib:
    def __init__(self, cdll):
        self.lib = cdll

    def frame(self, width, height):
        return Frame(self, width, height)

    def frame_nonfinite(self, frame_ptr, width, height, stream=None):
        """ezrt_frame_nonfinite: pixels of a device-resident RGBA32F frame with a non-finite R, G or B."""
        n = C.c_int64()
        if self.lib.ezrt_frame_nonfinite(C.c_void_p(frame_ptr), int(width), int(height), C.c_void_p(stream or 0),
                                         C.cast(C.byref(n), _abi.c_int64_p)) != 0:
            raise TraceError(self.lib.ezrt_last_error().decode())
        return int(n.value)

    def trim(self):
        """ezrt_trim: destroy the streams parked by destroyed scenes; returns how many."""
        return int(self.lib.ezrt_trim())

    def backend(self):
        return self.lib.ezrt_backend().decode()

    def scene_create(self, tri, nodes):
        tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 36)
        nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
        h = C.c_void_p()
        rc = self.lib.ezrt_scene_create(_fp(tri), tri.shape[0], _fp(nodes), nodes.shape[0], C.byref(h))
        if rc != 0:
            raise TraceError("%s (rc=%d)" % (self.lib.ezrt_last_error().decode(), rc))
        return Scene(self, h)

    def tonemap(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32).reshape(-1, 4)
        out = np.zeros((rgba.shape[0], 3), np.uint8)
        rc = self.lib.ezrt_tonemap(_fp(rgba), rgba.shape[0], out.ctypes.data_as(_abi.c_uint8_p))
        if rc != 0:
            raise TraceError(self.lib.ezrt_last_error().decode())
        return out

    def sobol(self, index0, n, n_dims=8):
        out = np.zeros((n, n_dims), np.float32)
        rc = self.lib.ezrt_sobol(int(index0), int(n), int(n_dims), _fp(out))
        if rc != 0:
            raise TraceError(self.lib.ezrt_last_error().decode())
        return out

    def debug_math(self, op, a, b=None, n=None):
        """ops 0-9: elementwise det-math; ops 10-12 (intersector audit): a = [n,6] rays,
        b = [n,6] boxes (10 hitAABB, 12 its v_min3/v_max3 form) or [n,9] triangles (11 hitTriangle);
        ops 13-16 (integrator 52's sampler, frame N = +z): b = [n,6] (roughness, anisotropic, metallic, clearcoat,
        clearcoatGloss, -), 13: a = [n,6] (V, L) -> pdf, 14/15/16: a = [n,6] (xi1, xi2, xi3, V) -> L.x / L.y / L.z."""
        a = np.ascontiguousarray(a, np.float32)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else np.zeros_like(a)
        n = a.size if n is None else int(n)
        out = np.zeros(n, np.float32) if op >= 10 else np.zeros_like(a)
        rc = self.lib.ezrt_debug_math(int(op), _fp(a), _fp(bb), n, _fp(out))
        if rc != 0:
            raise TraceError(self.lib.ezrt_last_error().decode())
        return out


def hip():
    """The product: hand-written gfx950 kernels behind the C ABI."""
    return TraceLib(_abi.load_hip())
