"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libezrt_ref_p{2,3,4,5}.so: the REFERENCE's
own host code compiled from /root/reference by oracle/ref_recipe/build_ref.py.

Only tests/ (and tests/golden/make_fixtures.py) import this.  On the GPU box /root/reference is
absent but the prebuilt .so files travel with the snapshot; `available()` says whether they exist.
Asset paths under /root/reference are only ever opened in this container (tests skip otherwise).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE = "/root/reference"
CHAPTER_DIR = {
    "p2": "part 2 -- BVH Accelerate Struct",
    "p3": "part 3 -- OpenGL Raytracing",
    "p4": "part 4 -- Disney Principle BRDF",
    "p5": "part 5 -- Importance Sampling & Low Discrepancy Sequence",
}
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)


def source_dir(tag):
    return os.path.join(REFERENCE, CHAPTER_DIR[tag], "source code")


def have_reference():
    return os.path.isdir(REFERENCE)


def lib_path(tag):
    return os.path.join(REF_DIR, "libezrt_ref_%s.so" % tag)


def available(tag="p3"):
    return os.path.exists(lib_path(tag))


def _fp(a):
    return a.ctypes.data_as(_F)


def _ip(a):
    return a.ctypes.data_as(_I)


_libs = {}


def _load(tag):
    if tag not in _libs:
        _libs[tag] = C.CDLL(lib_path(tag))
    return _libs[tag]


class Flat:
    """Chapters 3/4/5: flat-array scene build (P3/main.cpp:254-588), HDR loader, env cache."""

    def __init__(self, tag="p3"):
        self.tag = tag
        L = self.lib = _load(tag)
        L.ref_read_obj.argtypes = [C.c_char_p, _F, _F, C.c_int]
        L.ref_add_triangles.argtypes = [_F, C.c_int]
        L.ref_build.argtypes = [C.c_int, C.c_int]
        L.ref_build.restype = C.c_int
        L.ref_counts.argtypes = [_I, _I]
        L.ref_get_scene.argtypes = [_F, _I, _F]
        L.ref_get_transform_matrix.argtypes = [_F, _F, _F, _F]
        L.ref_material_default.argtypes = [_F]
        L.ref_hdr_load.argtypes = [C.c_char_p, _I, _I, C.POINTER(_F)]
        L.ref_hdr_load.restype = C.c_int
        L.ref_free.argtypes = [_F]
        L.ref_camera.argtypes = [C.c_float, C.c_float, C.c_float, _F, _F]
        L.ref_run_main.argtypes = [C.c_char_p, C.c_int]
        L.ref_run_main.restype = C.c_int
        L.ref_recorded_buffer_floats.restype = C.c_long
        L.ref_recorded_buffer_floats.argtypes = [C.c_int]
        L.ref_recorded_buffer_get.argtypes = [C.c_int, _F]
        L.ref_recorded_uniform_i.argtypes = [C.c_char_p]
        L.ref_recorded_uniform_i.restype = C.c_longlong
        if tag == "p5":
            L.ref_calculate_hdr_cache.argtypes = [_F, C.c_int, C.c_int]
            L.ref_calculate_hdr_cache.restype = _F

    def clear(self):
        self.lib.ref_scene_clear()

    def materialDefault(self):
        out = np.zeros(18, np.float32)
        self.lib.ref_material_default(_fp(out))
        return out

    def getTransformMatrix(self, r, t, s):
        out = np.zeros(16, np.float32)
        r, t, s = (np.asarray(x, np.float32) for x in (r, t, s))
        self.lib.ref_get_transform_matrix(_fp(r), _fp(t), _fp(s), _fp(out))
        return out

    def readObj(self, path, mat18, trans16, smooth):
        m = np.ascontiguousarray(mat18, np.float32)
        t = np.ascontiguousarray(trans16, np.float32)
        self.lib.ref_read_obj(str(path).encode(), _fp(m), _fp(t), int(smooth))

    def addTriangles(self, tri36):
        a = np.ascontiguousarray(tri36, np.float32).reshape(-1, 36)
        self.lib.ref_add_triangles(_fp(a), a.shape[0])

    def build(self, sah=True, leaf_n=8):
        return self.lib.ref_build(int(bool(sah)), int(leaf_n))

    def scene(self):
        """(tri [nt,36], node ints [nn,4] = left,right,n,index, node boxes [nn,6] = AA,BB)"""
        nt, nn = C.c_int(), C.c_int()
        self.lib.ref_counts(C.byref(nt), C.byref(nn))
        tri = np.zeros((nt.value, 36), np.float32)
        ints = np.zeros((nn.value, 4), np.int32)
        boxes = np.zeros((nn.value, 6), np.float32)
        self.lib.ref_get_scene(_fp(tri), _ip(ints), _fp(boxes))
        return tri, ints, boxes

    def encodedNodes(self):
        """the scene()'s nodes in the texel layout of the encode loop (P3/main.cpp:740-748)"""
        _, ints, boxes = self.scene()
        out = np.zeros((ints.shape[0], 12), np.float32)
        out[:, 0] = ints[:, 0]; out[:, 1] = ints[:, 1]
        out[:, 3] = ints[:, 2]; out[:, 4] = ints[:, 3]
        out[:, 6:12] = boxes
        return out

    def hdrLoad(self, path):
        w, h, p = C.c_int(), C.c_int(), _F()
        ok = self.lib.ref_hdr_load(str(path).encode(), C.byref(w), C.byref(h), C.byref(p))
        if not ok:
            return None
        try:
            return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
        finally:
            self.lib.ref_free(p)

    def calculateHdrCache(self, hdr):
        hdr = np.ascontiguousarray(hdr, np.float32)
        h, w, _ = hdr.shape
        p = self.lib.ref_calculate_hdr_cache(_fp(hdr), w, h)
        try:
            return np.ctypeslib.as_array(p, shape=(h, w, 3)).copy()
        finally:
            self.lib.ref_free(p)

    def camera(self, rotat, up, r):
        eye = np.zeros(3, np.float32)
        cam = np.zeros(16, np.float32)
        self.lib.ref_camera(float(rotat), float(up), float(r), _fp(eye), _fp(cam))
        return eye, cam

    def runMain(self, record_images=False):
        """the chapter's main() headless -> list of recorded GL_TEXTURE_BUFFER uploads"""
        rc = self.lib.ref_run_main(source_dir(self.tag).encode(), int(record_images))
        if rc != 0:
            raise RuntimeError("reference main() returned %d" % rc)
        out = []
        for i in range(self.lib.ref_recorded_buffers()):
            a = np.zeros(self.lib.ref_recorded_buffer_floats(i), np.float32)
            self.lib.ref_recorded_buffer_get(i, _fp(a))
            out.append(a)
        return out

    def uniformInt(self, name):
        return int(self.lib.ref_recorded_uniform_i(name.encode()))


class P2:
    """Chapter 2: pointer tree + the C++ twins of hitTriangle/hitAABB (P2/main.cpp:212-485)."""

    def __init__(self):
        L = self.lib = _load("p2")
        L.ref_p2_inf.restype = C.c_float
        L.ref_p2_set_triangles.argtypes = [_F, C.c_int]
        L.ref_p2_get_triangles.argtypes = [_F]
        L.ref_p2_build.argtypes = [C.c_int, C.c_int]
        L.ref_p2_get_tree.argtypes = [_I, _F]
        L.ref_p2_hit_triangle.argtypes = [_F, C.c_int, _F, _F]
        L.ref_p2_hit_aabb.argtypes = [_F, _F, C.c_int, _F]
        L.ref_p2_hit_triangle_array.argtypes = [_F, C.c_int, _I, _F]
        L.ref_p2_hit_bvh.argtypes = [_F, C.c_int, _I, _F]
        L.ref_p2_run_main.argtypes = [C.c_char_p]
        L.ref_p2_lines_get.argtypes = [_F]
        self.INF = L.ref_p2_inf()

    def setTriangles(self, tri9):
        a = np.ascontiguousarray(tri9, np.float32).reshape(-1, 9)
        self.lib.ref_p2_set_triangles(_fp(a), a.shape[0])

    def triangles(self):
        out = np.zeros((self.lib.ref_p2_count(), 9), np.float32)
        self.lib.ref_p2_get_triangles(_fp(out))
        return out

    def build(self, sah=True, leaf_n=8):
        nn = self.lib.ref_p2_build(int(bool(sah)), int(leaf_n))
        ints = np.zeros((nn, 4), np.int32)
        boxes = np.zeros((nn, 6), np.float32)
        self.lib.ref_p2_get_tree(_ip(ints), _fp(boxes))
        return ints, boxes

    def hitTriangle(self, tri9, rays):
        tri9 = np.ascontiguousarray(tri9, np.float32).reshape(-1, 9)
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        t = np.zeros(tri9.shape[0], np.float32)
        self.lib.ref_p2_hit_triangle(_fp(tri9), tri9.shape[0], _fp(rays), _fp(t))
        return t

    def hitAABB(self, rays, boxes6):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        boxes6 = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        t = np.zeros(rays.shape[0], np.float32)
        self.lib.ref_p2_hit_aabb(_fp(rays), _fp(boxes6), rays.shape[0], _fp(t))
        return t

    def hitTriangleArray(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        idx = np.zeros(rays.shape[0], np.int32)
        t = np.zeros(rays.shape[0], np.float32)
        self.lib.ref_p2_hit_triangle_array(_fp(rays), rays.shape[0], _ip(idx), _fp(t))
        return idx, t

    def hitBVH(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        idx = np.zeros(rays.shape[0], np.int32)
        t = np.zeros(rays.shape[0], np.float32)
        self.lib.ref_p2_hit_bvh(_fp(rays), rays.shape[0], _ip(idx), _fp(t))
        return idx, t

    def runMain(self):
        rc = self.lib.ref_p2_run_main(source_dir("p2").encode())
        if rc != 0:
            raise RuntimeError("reference main() returned %d" % rc)
        lines = np.zeros((self.lib.ref_p2_lines_count(), 3), np.float32)
        self.lib.ref_p2_lines_get(_fp(lines))
        return lines


# ----------------------------------------------------------------------------------------------
# the chapters' FRAGMENT SHADERS compiled (oracle/ref_recipe/wrap_fsh.cpp, fsh_pass.py, shim/glsl_shim.h)
def fsh_available(chapter=5):
    return os.path.exists(os.path.join(REF_DIR, "libezrt_ref_fsh_p%d.so" % chapter))


class Fsh:
    """One chapter's shaders/fshader.fsh as a shared library: main() per pixel-sample, and its functions."""

    OUT_WIDTH = {1: 3, 2: 3, 3: 3, 4: 1, 5: 1, 6: 3, 7: 3, 8: 12, 9: 3}
    IN_WIDTH = {1: 9, 2: 9, 3: 9, 4: 9, 5: 3, 6: 2, 7: 3, 8: 6, 9: 5}

    def __init__(self, chapter):
        self.chapter = int(chapter)
        L = self.lib = C.CDLL(os.path.join(REF_DIR, "libezrt_ref_fsh_p%d.so" % self.chapter))
        L.fsh_set_scene.argtypes = [_F, C.c_int, _F, C.c_int]
        L.fsh_set_env.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int]
        L.fsh_set_camera.argtypes = [_F, _F, C.c_int, C.c_int]
        L.fsh_set_integrator.argtypes = [C.c_int, C.c_int]
        L.fsh_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, _F]
        L.fsh_seed.argtypes = [C.c_int, C.c_int, C.c_uint]
        L.fsh_seed.restype = C.c_uint
        L.fsh_fn.argtypes = [C.c_int, _F, _F, C.c_int, _F]
        L.fsh_fn.restype = C.c_int
        assert L.fsh_chapter() == self.chapter

    def set_scene(self, tri, nodes):
        tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 36)
        nodes = np.ascontiguousarray(nodes, np.float32).reshape(-1, 12)
        self.lib.fsh_set_scene(_fp(tri), tri.shape[0], _fp(nodes), nodes.shape[0])

    def set_env(self, hdr, cache, bilinear):
        hdr = np.ascontiguousarray(hdr, np.float32)
        h, w, _ = hdr.shape
        cp = None
        if cache is not None:
            cache = np.ascontiguousarray(cache, np.float32)
            cp = _fp(cache)
        self.lib.fsh_set_env(_fp(hdr), cp, w, h, int(bilinear))

    def set_camera(self, eye, camera_rotate, width, height):
        e = np.ascontiguousarray(eye, np.float32)
        c = np.ascontiguousarray(camera_rotate, np.float32).reshape(16)
        self.lib.fsh_set_camera(_fp(e), _fp(c), int(width), int(height))
        self.width, self.height = int(width), int(height)

    def set_integrator(self, max_bounce, use_importance_sampling):
        self.lib.fsh_set_integrator(int(max_bounce), int(use_importance_sampling))

    def render(self, frame0, spp, accum=None, rect=None):
        if accum is None:
            accum = np.zeros((self.height, self.width, 4), np.float32)
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.width, self.height)
        self.lib.fsh_render(x0, y0, x1, y1, int(frame0), int(spp), _fp(accum))
        return accum

    def seed(self, ix, iy, frame):
        return int(self.lib.fsh_seed(int(ix), int(iy), int(frame)))

    def fn(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, self.IN_WIDTH[op])
        n = a.shape[0]
        bb = np.ascontiguousarray(b, np.float32) if b is not None else np.zeros((n, 18), np.float32)
        out = np.zeros((n, self.OUT_WIDTH[op]), np.float32)
        rc = self.lib.fsh_fn(int(op), _fp(a), _fp(bb), n, _fp(out))
        if rc != 0:
            raise ValueError("chapter %d's shader has no op %d" % (self.chapter, op))
        return out
