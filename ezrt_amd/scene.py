"""Host scene build: Python mirror of the reference's pre-frame sequence.

Thin wrappers over libezrt_scene.so (C++; include/ezrt_scene.hpp).  Names follow
the reference: Material, readObj, getTransformMatrix, buildBVH,
buildBVHwithSAH, encode loops, HDRLoader.load, calculateHdrCache
(P3/main.cpp:28-57, 254-588, 720-748; P5/main.cpp:592-689).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi

_F = _abi.c_float_p


def _fp(a):
    return a.ctypes.data_as(_F)


def _check(rc, lib):
    if rc != 0:
        raise RuntimeError("ezrt host: %s" % lib.ezrt_host_last_error().decode())


@dataclass
class Material:
    """P3/main.cpp:28-43 (defaults of chapter 3); `Material.disney()` = P4/P5 defaults."""
    emissive: tuple = (0.0, 0.0, 0.0)
    baseColor: tuple = (1.0, 1.0, 1.0)
    subsurface: float = 0.0
    metallic: float = 0.0
    specular: float = 0.0
    specularTint: float = 0.0
    roughness: float = 0.0
    anisotropic: float = 0.0
    sheen: float = 0.0
    sheenTint: float = 0.0
    clearcoat: float = 0.0
    clearcoatGloss: float = 0.0
    IOR: float = 1.0
    transmission: float = 0.0

    @staticmethod
    def disney(**kw):
        m = Material(specular=0.5, roughness=0.5, sheenTint=0.5, clearcoatGloss=1.0)
        for k, v in kw.items():
            setattr(m, k, v)
        return m

    def to18(self):
        return np.array(list(self.emissive) + list(self.baseColor) + [
            self.subsurface, self.metallic, self.specular, self.specularTint, self.roughness, self.anisotropic,
            self.sheen, self.sheenTint, self.clearcoat, self.clearcoatGloss, self.IOR, self.transmission],
            dtype=np.float32)


def getTransformMatrix(rotateCtrl, translateCtrl, scaleCtrl):
    """P3/main.cpp:254-270 -> column-major mat4 as float32[16]."""
    lib = _abi.load_host()
    out = np.zeros(16, np.float32)
    r = np.asarray(rotateCtrl, np.float32)
    t = np.asarray(translateCtrl, np.float32)
    s = np.asarray(scaleCtrl, np.float32)
    _check(lib.ezrt_host_get_transform_matrix(_fp(r), _fp(t), _fp(s), _fp(out)), lib)
    return out


def setTieOrder(library_sort):
    """False (default): equal centroid keys keep their order (stable; == the GPU builder).  True: this
    toolchain's std::sort, i.e. the permutation the reference's std::sort produces when built here."""
    lib = _abi.load_host()
    _check(lib.ezrt_host_set_tie_order(int(bool(library_sort))), lib)


class HostScene:
    """std::vector<Triangle> + std::vector<BVHNode> of a reference main()."""

    def __init__(self):
        self._lib = _abi.load_host()
        self._h = self._lib.ezrt_host_scene_new()
        if not self._h:
            raise MemoryError("ezrt_host_scene_new failed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ezrt_host_scene_free(self._h)
            self._h = None

    def readObj(self, filepath, material, trans, smoothNormal):
        m = material.to18()
        t = np.ascontiguousarray(trans, np.float32)
        _check(self._lib.ezrt_host_read_obj(self._h, str(filepath).encode(), _fp(m), _fp(t), int(smoothNormal)),
               self._lib)

    def readObjText(self, text, material, trans, smoothNormal):
        if isinstance(text, str):
            text = text.encode()
        m = material.to18()
        t = np.ascontiguousarray(trans, np.float32)
        _check(self._lib.ezrt_host_read_obj_text(self._h, text, len(text), _fp(m), _fp(t), int(smoothNormal)),
               self._lib)

    def addTriangles(self, tri36):
        a = np.ascontiguousarray(tri36, np.float32).reshape(-1, 36)
        _check(self._lib.ezrt_host_add_triangles(self._h, _fp(a), a.shape[0]), self._lib)

    def buildBVH(self, n=8):
        _check(self._lib.ezrt_host_build_bvh(self._h, 0, n), self._lib)

    def buildBVHwithSAH(self, n=8):
        _check(self._lib.ezrt_host_build_bvh(self._h, 1, n), self._lib)

    def buildStats(self):
        out = (C.c_int64 * 3)()
        _check(self._lib.ezrt_host_build_stats(self._h, out), self._lib)
        return {"inf_cap_nodes": out[0], "sorts": out[1], "max_depth": out[2]}

    def counts(self):
        a, b = C.c_int(), C.c_int()
        _check(self._lib.ezrt_host_counts(self._h, C.byref(a), C.byref(b)), self._lib)
        return a.value, b.value

    def encode(self):
        """The two flat float arrays the trace consumes (P3/main.cpp:720-748)."""
        nt, nn = self.counts()
        tri = np.zeros((nt, 36), np.float32)
        nodes = np.zeros((nn, 12), np.float32)
        _check(self._lib.ezrt_host_encode(self._h, _fp(tri), _fp(nodes)), self._lib)
        return tri, nodes


def hdrLoad(path=None, data=None):
    """HDRLoader::load -> float32 [h, w, 3] (row 0 = top scanline)."""
    lib = _abi.load_host()
    w, h = C.c_int(), C.c_int()
    p = _F()
    if data is not None:
        _check(lib.ezrt_host_hdr_load_memory(bytes(data), len(data), C.byref(w), C.byref(h), C.byref(p)), lib)
    else:
        _check(lib.ezrt_host_hdr_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(p)), lib)
    try:
        arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.ezrt_host_free(p)
    return arr


def calculateHdrCache(hdr):
    """P5/main.cpp:592-689 -> float32 [h, w, 3] = (x/w, y/h, pdf)."""
    lib = _abi.load_host()
    hdr = np.ascontiguousarray(hdr, np.float32)
    h, w, _ = hdr.shape
    out = np.zeros_like(hdr)
    _check(lib.ezrt_host_hdr_cache(_fp(hdr), w, h, _fp(out)), lib)
    return out


def camera(rotatAngle=0.0, upAngle=0.0, r=4.0):
    """eye, cameraRotate of display() (P3/main.cpp:607-610; defaults 148-150)."""
    lib = _abi.load_host()
    eye = np.zeros(3, np.float32)
    m = np.zeros(16, np.float32)
    _check(lib.ezrt_host_camera(float(rotatAngle), float(upAngle), float(r), _fp(eye), _fp(m)), lib)
    return eye, m


def p2Query(tri9, rays, sah=True, leaf_n=8, use_bvh=True):
    """Chapter 2's CPU query (ezrt::p2, P2/main.cpp:242-485): build the pointer tree over
    `tri9` [n, 9], shoot `rays` [m, 6] through hitBVH (or hitTriangleArray over everything when
    use_bvh is False).  Returns (sorted triangles [n, 9], hit index [m] into them or -1,
    distance [m], INF = 114514 on a miss)."""
    lib = _abi.load_host()
    tri9 = np.ascontiguousarray(tri9, np.float32).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    out = np.zeros_like(tri9)
    idx = np.zeros(rays.shape[0], np.int32)
    t = np.zeros(rays.shape[0], np.float32)
    _check(lib.ezrt_host_p2_query(_fp(tri9), tri9.shape[0], int(bool(sah)), int(leaf_n), _fp(rays), rays.shape[0],
                                  int(bool(use_bvh)), _fp(out), idx.ctypes.data_as(C.POINTER(C.c_int)), _fp(t)), lib)
    return out, idx, t
