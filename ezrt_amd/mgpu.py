"""numpy-facing wrapper over include/ezrt_mgpu.h: one host process, N devices, one frame.

    m = Mgpu(trace.hip(), tri, nodes, devices=[0, 1, 2, 3])          # RCCL over xGMI by default
    m.set_env(hdr, cache); m.render(params); frame = m.gather()

`devices` may repeat an ordinal (several shards on one GPU) with transport="peer" or "host".  No fallback: the
library is whatever TraceLib was opened on (libezrt_hip.so in the product; tests also drive the CPU oracle's
implementation of the same header)."""
import ctypes as C

import numpy as np

from . import _abi
from .trace import TraceError, _fp

TRANSPORTS = {"rccl": _abi.TRANSPORT_RCCL, "peer": _abi.TRANSPORT_PEER, "host": _abi.TRANSPORT_HOST}


class Mgpu:
    def __init__(self, tracelib, tri, nodes, devices, transport="rccl"):
        self._lib = tracelib.lib
        tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 36)
        nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
        self.devices = [int(d) for d in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = self._lib.ezrt_mgpu_create(_fp(tri), tri.shape[0], _fp(nodes), nodes.shape[0], dev, len(self.devices),
                                        TRANSPORTS[transport], C.byref(h))
        if rc != 0:
            raise TraceError("%s (rc=%d)" % (self._lib.ezrt_last_error().decode(), rc))
        self._h = h
        self._shape = None

    def close(self):
        if self._h:
            self._lib.ezrt_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise TraceError("%s (rc=%d)" % (self._lib.ezrt_last_error().decode(), rc))

    def set_env(self, hdr, cache=None, filter=_abi.FILTER_BILINEAR):
        hdr = np.ascontiguousarray(hdr, np.float32)
        h, w, _ = hdr.shape
        cp = None
        if cache is not None:
            cache = np.ascontiguousarray(cache, np.float32)
            cp = _fp(cache)
        self._ck(self._lib.ezrt_mgpu_set_env(self._h, _fp(hdr), cp, w, h, int(filter)))

    def set_sampler(self, sobol_dims):
        self._ck(self._lib.ezrt_mgpu_set_sampler(self._h, int(sobol_dims)))

    def set_option(self, name, value):
        self._ck(self._lib.ezrt_mgpu_set_option(self._h, name.encode(), int(value)))

    def render(self, params):
        self._ck(self._lib.ezrt_mgpu_render(self._h, C.byref(params)))
        self._shape = (int(params.height), int(params.width), 4)

    def gather(self, to_host=True):
        """Close the frame; returns the assembled [H, W, 4] running mean (or None with to_host=False: the frame
        stays on the root device, see frame_device())."""
        if not to_host or self._shape is None:
            self._ck(self._lib.ezrt_mgpu_gather(self._h, None))
            return None
        out = np.zeros(self._shape, np.float32)
        self._ck(self._lib.ezrt_mgpu_gather(self._h, _fp(out)))
        return out

    def frame_device(self):
        p = C.c_void_p()
        self._ck(self._lib.ezrt_mgpu_frame_device(self._h, C.byref(p)))
        return p.value

    def counters(self):
        out = (C.c_uint64 * _abi.EZRT_CTR_COUNT)()
        self._ck(self._lib.ezrt_mgpu_counters(self._h, out))
        return dict(zip(_abi.CTR_NAMES, [int(x) for x in out]))

    def last_ms(self):
        r = np.zeros(len(self.devices), np.float32)
        g = C.c_float()
        b = C.c_int64()
        self._ck(self._lib.ezrt_mgpu_last_ms(self._h, _fp(r), C.cast(C.byref(g), _abi.c_float_p), C.cast(C.byref(b), _abi.c_int64_p)))
        return {"render_ms": [float(x) for x in r], "gather_ms": float(g.value), "gather_bytes": int(b.value)}
