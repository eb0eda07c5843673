"""Canonical scenes of the BASELINE.json configs, rebuilt through
the host scene-build API exactly as the reference's main() would:
readObj x N -> nodes = {testNode} -> buildBVHwithSAH(.., 8) -> encode.

Geometry comes from ezrt_amd/assets/meshes.npz (raw OBJ vertices/faces extracted
from the reference's model files by tests/golden/make_fixtures.py) turned back
into OBJ text, so readObj's parsing + normalisation path is exercised.  Env maps:
`shipped_hdr()` = the RGBE texels of the only HDR the reference ships
(P4/HDR/peppermint_powerplant_4k.hdr, SURVEY-C2's env; a data asset under
ezrt_amd/assets/, decoded with HDRLoader's formula), or the procedural
`synthetic_hdr` (no asset needed).  /root/reference is never read at run time.
"""
import os

import numpy as np

from . import scene as S
from ._abi import (FILTER_BILINEAR, FILTER_NEAREST, INTEGRATOR_P3_DIFFUSE, INTEGRATOR_P4_DISNEY,
                   INTEGRATOR_P5_MIS, INTEGRATOR_P5_SOBOL)

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
MESHES = os.path.join(ASSETS, "meshes.npz")
SHIPPED_HDR = os.path.join(ASSETS, "env_peppermint_powerplant_1024x512_rgbe.npz")

_mesh_cache = {}


def mesh(name):
    if not _mesh_cache:
        with np.load(MESHES) as z:
            for k in z.files:
                _mesh_cache[k] = z[k]
    return _mesh_cache[name + "_v"], _mesh_cache[name + "_f"]


def obj_text(v, f):
    """OBJ text whose strtof parse returns exactly the float32 vertices `v`."""
    lines = ["v %.9g %.9g %.9g" % (float(a), float(b), float(c)) for a, b, c in v]
    lines += ["f %d %d %d" % (a + 1, b + 1, c + 1) for a, b, c in f]
    return ("\n".join(lines) + "\n").encode()


def subdivide(v, f, levels=1):
    """1->4 midpoint subdivision in float32 (shared edge midpoints are shared vertices)."""
    v = np.asarray(v, np.float32)
    f = np.asarray(f, np.int64)
    for _ in range(levels):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * (v.shape[0] + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a = (uniq // (v.shape[0] + 1)).astype(np.int64)
        b = (uniq % (v.shape[0] + 1)).astype(np.int64)
        mid = ((v[a] + v[b]) * np.float32(0.5)).astype(np.float32)
        n0 = v.shape[0]
        nf = f.shape[0]
        m01 = n0 + inv[0:nf]
        m12 = n0 + inv[nf:2 * nf]
        m20 = n0 + inv[2 * nf:3 * nf]
        v = np.concatenate([v, mid], axis=0)
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], 1),
            np.stack([m01, f[:, 1], m12], 1),
            np.stack([m20, m12, f[:, 2]], 1),
            np.stack([m01, m12, m20], 1)], axis=0)
    return v, f.astype(np.int32)


_shipped_hdr_cache = {}


def shipped_hdr():
    """float32 [512, 1024, 3], row 0 = top scanline: the reference's only shipped env map
    (P4/HDR/peppermint_powerplant_4k.hdr) from its RGBE texels, converted as HDRLoader does
    (lib/hdrloader.cpp:97-114: component / 256 * 2^(E - 128), exact in fp32).  Bit-identical to the
    reference loader's output on the file (tests/test_ref_pin.py, tests/golden/hdr_probe.json)."""
    if "a" not in _shipped_hdr_cache:
        with np.load(SHIPPED_HDR) as z:
            rgbe = z["rgbe"]
        v = rgbe[..., :3].astype(np.float32) / np.float32(256.0)
        d = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 128).astype(np.float32)
        _shipped_hdr_cache["a"] = np.ascontiguousarray(v * d[..., None], np.float32)
    return _shipped_hdr_cache["a"]


def env_map(name):
    """"shipped" | "synthetic" | an array | None -> float32 [h, w, 3] or None"""
    if isinstance(name, str):
        if name == "shipped":
            return shipped_hdr()
        if name == "synthetic":
            return synthetic_hdr()
        raise ValueError("unknown env map %r" % name)
    return name


def _hash_u32(x):
    x = np.asarray(x, np.uint32)
    x = (x ^ np.uint32(61)) ^ (x >> np.uint32(16))
    x = x * np.uint32(9)
    x = x ^ (x >> np.uint32(4))
    x = x * np.uint32(0x27d4eb2d)
    x = x ^ (x >> np.uint32(15))
    return x


def synthetic_hdr(w=1024, h=512):
    """Deterministic procedural equirect env (float32 [h, w, 3], row 0 = top): sky gradient,
    ground, a hot sun disc, three bright panels, +-5 % integer-hash grain.  Only IEEE + - * /
    on float32, so it is bit-identical on every machine."""
    f32 = np.float32
    rows = ((np.arange(h, dtype=f32) + f32(0.5)) / f32(h))[:, None]
    cols = ((np.arange(w, dtype=f32) + f32(0.5)) / f32(w))[None, :]
    up = np.clip((f32(0.5) - rows) * f32(2.0), f32(0), f32(1))       # 1 at zenith .. 0 at horizon
    dn = np.clip((rows - f32(0.5)) * f32(2.0), f32(0), f32(1))
    sky = np.stack([f32(0.85) - f32(0.55) * up, f32(0.9) - f32(0.4) * up, f32(1.0) - f32(0.1) * up], -1)
    gnd = np.stack([f32(0.30) - f32(0.1) * dn, f32(0.27) - f32(0.1) * dn, f32(0.22) - f32(0.08) * dn], -1)
    img = np.where((rows < f32(0.5))[..., None], sky + np.zeros((h, w, 3), f32), gnd + np.zeros((h, w, 3), f32))
    img = img.astype(f32)
    # sun
    du = (cols - f32(0.30)) * f32(2.0)
    dv = rows - f32(0.28)
    d2 = (du * du + dv * dv) / f32(0.0016)
    k = np.clip(f32(1.0) - d2, f32(0), f32(1))
    sun = (k * k) * f32(90.0)
    img = img + sun[..., None] * np.array([1.0, 0.92, 0.75], f32)
    # panels
    for (u0, u1, v0, v1, col) in ((0.55, 0.62, 0.30, 0.42, (6.0, 6.5, 7.0)), (0.80, 0.90, 0.36, 0.40, (12.0, 9.0, 5.0)),
                                  (0.05, 0.09, 0.20, 0.45, (3.0, 4.0, 3.5))):
        m = (cols > f32(u0)) & (cols < f32(u1)) & (rows > f32(v0)) & (rows < f32(v1))
        img = np.where(m[..., None], np.array(col, f32)[None, None, :], img)
    idx = (np.arange(h, dtype=np.uint32)[:, None] * np.uint32(w) + np.arange(w, dtype=np.uint32)[None, :])
    g = (_hash_u32(idx) >> np.uint32(8)).astype(f32) / f32(16777216.0)
    img = img * (f32(0.95) + f32(0.1) * g)[..., None]
    return np.ascontiguousarray(img.astype(f32))


class BuiltScene:
    def __init__(self, name, tri, nodes, build_stats, hdr=None, cache=None, env_filter=FILTER_BILINEAR):
        self.name = name
        self.tri = tri
        self.nodes = nodes
        self.build_stats = build_stats
        self.hdr = hdr
        self.cache = cache
        self.env_filter = env_filter

    def upload(self, tracelib):
        s = tracelib.scene_create(self.tri, self.nodes)
        if self.hdr is not None:
            s.set_env(self.hdr, self.cache, self.env_filter)
        return s


# buildBVHwithSAH on the GPU (ezrt_build_sah: the SAME arrays as the host builder, bit for bit -- tests/test_gpu_lbvh.py --
# ~70x faster at 10^6 triangles): the default for scenes of >= GPU_BUILD_MIN_TRIS triangles when a GPU is visible
# (round 4); EZRT_GPU_BUILD=1 / 0 forces it on / off for every scene, gpu_build=True / False per call.
GPU_BUILD_MIN_TRIS = 100_000


def _auto_gpu_build(n_tri):
    env = os.environ.get("EZRT_GPU_BUILD", "")
    if env != "":
        return env != "0"
    if n_tri < GPU_BUILD_MIN_TRIS:
        return False
    from . import build
    try:
        return build.device_count() > 0
    except (RuntimeError, OSError):
        # libezrt_hip.so or its ROCm dependencies cannot be loaded on this host: the host builder it is (oracle-only and
        # CPU flows -- golden generation, CPU tests on C3 / C5 -- must not need the HIP library)
        return False


def _finish(name, hs, leaf_n, hdr, want_cache, env_filter, sah=True, gpu_build=None):
    if gpu_build is None:
        gpu_build = sah and _auto_gpu_build(hs.counts()[0])
    cache = S.calculateHdrCache(hdr) if (want_cache and hdr is not None) else None
    if sah and gpu_build:
        from . import build
        raw, _ = hs.encode()
        tri, nodes, ms = build.build_sah(raw, leaf_n)
        return BuiltScene(name, tri, nodes, {"gpu_build_ms": ms}, hdr, cache, env_filter)
    if sah:
        hs.buildBVHwithSAH(leaf_n)
    else:
        hs.buildBVH(leaf_n)
    tri, nodes = hs.encode()
    return BuiltScene(name, tri, nodes, hs.buildStats(), hdr, cache, env_filter)


def bunny_scene(subdiv=0, materials="p4", hdr="synthetic", want_cache=False, env_filter=FILTER_BILINEAR, leaf_n=8,
                sah=True):
    """The P3 scene (P3/main.cpp:690-701): smooth Bunny T=(0.3,-1.6,0) S=1.5, floor box
    S=(18.83,0.01,18.83) T=(0,-1.4,0), emissive (30,20,10) sphere at (0,0.9,0).
    subdiv=2 gives the "~70k" variant (79 488 + 332 triangles).  materials: "p3" = chapter-3
    Material defaults, "p4" = chapter-4/5 defaults."""
    mk = (lambda **kw: S.Material(**kw)) if materials == "p3" else (lambda **kw: S.Material.disney(**kw))
    hs = S.HostScene()
    bv, bf = mesh("bunny")
    if subdiv:
        bv, bf = subdivide(bv, bf, subdiv)
    hs.readObjText(obj_text(bv, bf), mk(baseColor=(1, 1, 1)), S.getTransformMatrix((0, 0, 0), (0.3, -1.6, 0), (1.5, 1.5, 1.5)), True)
    qv, qf = mesh("quad")
    hs.readObjText(obj_text(qv, qf), mk(baseColor=(0.725, 0.71, 0.68)),
                   S.getTransformMatrix((0, 0, 0), (0, -1.4, 0), (18.83, 0.01, 18.83)), False)
    sv, sf = mesh("sphere")
    hs.readObjText(obj_text(sv, sf), mk(baseColor=(1, 1, 1), emissive=(30, 20, 10)),
                   S.getTransformMatrix((0, 0, 0), (0.0, 0.9, -0.0), (1, 1, 1)), False)
    h = env_map(hdr)
    return _finish("bunny_sub%d" % subdiv, hs, leaf_n, h, want_cache, env_filter, sah)


def p5_scene(subdiv=0, hdr="synthetic", leaf_n=8):
    """The P5 scene with the Bunny standing in for the missing teapot.obj
    (P5/main.cpp:795-819): metallic clear-coated gold body, near-mirror floor."""
    hs = S.HostScene()
    bv, bf = mesh("bunny")
    if subdiv:
        bv, bf = subdivide(bv, bf, subdiv)
    m = S.Material.disney(roughness=0.5, specular=1.0, metallic=1.0, clearcoat=1.0, clearcoatGloss=0.0,
                          baseColor=(1, 0.73, 0.25))
    hs.readObjText(obj_text(bv, bf), m, S.getTransformMatrix((0, 0, 0), (0.1, -1.0, 0), (0.75, 0.75, 0.75)), True)
    m = S.Material.disney(roughness=0.01, metallic=0.1, specular=1.0, clearcoat=1.0, clearcoatGloss=0.0,
                          baseColor=(1, 1, 1))
    qv, qf = mesh("quad")
    hs.readObjText(obj_text(qv, qf), m, S.getTransformMatrix((0, 0, 0), (0, -0.5, 0), (13000.0, 0.01, 13000.0)), False)
    h = env_map(hdr)
    return _finish("p5_sub%d" % subdiv, hs, leaf_n, h, True, FILTER_BILINEAR)


def cornell_scene(leaf_n=8):
    """The 12-triangle Cornell box of part 1 (P1/main.cpp:338-360): 10 wall triangles + 2 light
    triangles, as a flat triangle list through addTriangles (no OBJ)."""
    RED, GREEN, WHITE = (1, 0.5, 0.5), (0.5, 1, 0.5), (1, 1, 1)

    def tri(p1, p2, p3, col, emissive=(0, 0, 0)):
        p1, p2, p3 = (np.array(p, np.float32) for p in (p1, p2, p3))
        n = np.cross(p2 - p1, p3 - p1).astype(np.float32)
        n = n / np.float32(np.sqrt(np.float32(n @ n)))
        mat = S.Material.disney(baseColor=col, emissive=emissive).to18()
        return np.concatenate([p1, p2, p3, n, n, n, mat]).astype(np.float32)

    T = []
    # light
    T.append(tri((0.4, 0.99, 0.4), (-0.4, 0.99, -0.4), (-0.4, 0.99, 0.4), WHITE, (12, 12, 12)))
    T.append(tri((0.4, 0.99, 0.4), (0.4, 0.99, -0.4), (-0.4, 0.99, -0.4), WHITE, (12, 12, 12)))
    # bottom, top, back
    T.append(tri((1, -1, 1), (-1, -1, -1), (-1, -1, 1), WHITE))
    T.append(tri((1, -1, 1), (1, -1, -1), (-1, -1, -1), WHITE))
    T.append(tri((1, 1, 1), (-1, 1, 1), (-1, 1, -1), WHITE))
    T.append(tri((1, 1, 1), (-1, 1, -1), (1, 1, -1), WHITE))
    T.append(tri((1, -1, -1), (-1, 1, -1), (-1, -1, -1), WHITE))
    T.append(tri((1, -1, -1), (1, 1, -1), (-1, 1, -1), WHITE))
    # left, right
    T.append(tri((-1, -1, -1), (-1, 1, 1), (-1, -1, 1), RED))
    T.append(tri((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), RED))
    T.append(tri((1, 1, 1), (1, -1, -1), (1, -1, 1), GREEN))
    T.append(tri((1, -1, -1), (1, 1, 1), (1, 1, -1), GREEN))
    hs = S.HostScene()
    hs.addTriangles(np.stack(T))
    return _finish("cornell", hs, leaf_n, None, False, FILTER_NEAREST)


# (width, height, spp, max_bounce, integrator, camera(rot, up, r)) of the BASELINE.json configs
CONFIGS = {
    "C1": dict(width=256, height=256, spp=1, max_bounce=4, integrator=INTEGRATOR_P3_DIFFUSE, camera=(0, 0, 4)),
    "C2": dict(width=512, height=512, spp=64, max_bounce=4, integrator=INTEGRATOR_P5_SOBOL, camera=(0, 0, 4)),
    "C3": dict(width=1024, height=1024, spp=128, max_bounce=4, integrator=INTEGRATOR_P4_DISNEY, camera=(0, 15, 8)),
    "C4": dict(width=1024, height=1024, spp=256, max_bounce=2, integrator=INTEGRATOR_P5_MIS, camera=(90, 10, 2)),
    "C5": dict(width=2048, height=2048, spp=512, max_bounce=8, integrator=INTEGRATOR_P5_MIS, camera=(30, 25, 10)),
}


def _unit(seed, k):
    """k-th deterministic float32 in [0,1) of instance `seed` (Wang hash, as the shader's RNG)."""
    h = int(_hash_u32(np.array([seed * 9781 + k * 6271 + 1], np.uint32))[0])
    return np.float32(h >> 8) / np.float32(16777216.0)


def mega_scene(hdr="synthetic", leaf_n=8, gpu_build=None):
    """C5: exactly 1 000 000 triangles -- 48 instances of the 20 480-face sphere (= the face count of
    the reference's sphere2.obj) with per-instance transform and Disney parameters from
    wang_hash(instance id), 3 Bunnies (14 904), and 1 028 two-triangle quads: a 32x32 tiled floor
    plus 4 emissive panels."""
    hs = S.HostScene()
    sv, sf = mesh("sphere")
    sv, sf = subdivide(sv, sf, 3)
    c = sv.mean(axis=0, dtype=np.float32)
    r = np.sqrt(((sv[:162] - c) ** 2).sum(1)).mean(dtype=np.float32)
    dv = sv - c
    ln = np.sqrt((dv * dv).sum(1, dtype=np.float32)).astype(np.float32)
    sv = (c + dv * (r / ln)[:, None]).astype(np.float32)
    stext = obj_text(sv, sf)
    for i in range(48):
        gx, gz = i % 8, i // 8
        s = 0.55 + 0.5 * float(_unit(i, 0))
        pos = (-5.6 + 1.6 * gx + 0.5 * (float(_unit(i, 1)) - 0.5), -1.4 + s * 0.5 + 1.5 * float(_unit(i, 2)) * (i % 3 == 0),
               -4.0 + 1.6 * gz + 0.5 * (float(_unit(i, 3)) - 0.5))
        m = S.Material.disney(baseColor=(0.3 + 0.7 * float(_unit(i, 4)), 0.3 + 0.7 * float(_unit(i, 5)),
                                         0.3 + 0.7 * float(_unit(i, 6))),
                              metallic=float(_unit(i, 7)), roughness=0.05 + 0.75 * float(_unit(i, 8)),
                              clearcoat=float(_unit(i, 9)), subsurface=0.5 * float(_unit(i, 10)))
        rot = (360.0 * float(_unit(i, 11)), 360.0 * float(_unit(i, 12)), 0.0)
        hs.readObjText(stext, m, S.getTransformMatrix(rot, pos, (s, s, s)), True)
    bv, bf = mesh("bunny")
    btext = obj_text(bv, bf)
    gold = S.Material.disney(roughness=0.5, specular=1.0, metallic=1.0, clearcoat=1.0, clearcoatGloss=0.0,
                             baseColor=(1, 0.73, 0.25))
    for k, (x, z, ry) in enumerate(((-2.0, 4.4, 20.0), (0.5, 4.8, -35.0), (3.0, 4.2, 140.0))):
        hs.readObjText(btext, gold, S.getTransformMatrix((0, ry, 0), (x, -1.55, z), (1.4, 1.4, 1.4)), True)
    quad = obj_text(np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32),
                    np.array([[0, 2, 1], [0, 3, 2]], np.int32))
    for i in range(1024):
        tx, tz = i % 32, i // 32
        shade = 0.55 + 0.25 * ((tx + tz) & 1)
        m = S.Material.disney(baseColor=(shade, shade, shade), roughness=0.2 + 0.6 * float(_unit(1000 + i, 0)))
        hs.readObjText(quad, m, S.getTransformMatrix((0, 0, 0), (-15.5 + tx, -1.4, -15.5 + tz), (1.0, 1.0, 1.0)), False)
    for k, (x, z) in enumerate(((-4.0, -2.0), (4.0, -2.0), (-4.0, 3.0), (4.0, 3.0))):
        m = S.Material.disney(baseColor=(1, 1, 1), emissive=(18.0, 16.0 - 2.0 * k, 10.0 + 2.0 * k))
        hs.readObjText(quad, m, S.getTransformMatrix((180, 0, 0), (x, 3.2, z), (1.5, 1.0, 1.5)), False)
    h = env_map(hdr)
    return _finish("mega_1m", hs, leaf_n, h, True, FILTER_BILINEAR, gpu_build=gpu_build)


def disney_grid_scene(subdiv=3, hdr="synthetic", leaf_n=8):
    """C3: 5x5 grid of spheres (sphere.obj subdivided `subdiv` times: 3 -> 20 480 faces each, the
    face count of the reference's sphere2.obj), metallic in {0,.25,.5,.75,1} x roughness in
    {.1,.2,.3,.5,.8}, baseColor (0.75,0.7,0.15), clearcoat 1, over a floor box -- the sweeps of
    T4 tutorial.md:487-491 / P4/main.cpp:696-714 extended to a grid.  NEAREST env (P4 filter)."""
    hs = S.HostScene()
    sv, sf = mesh("sphere")
    if subdiv:
        sv, sf = subdivide(sv, sf, subdiv)
        # push the new vertices back onto the sphere (float32): midpoint subdivision alone keeps facets
        c = sv.mean(axis=0, dtype=np.float32)
        r = np.sqrt(((sv[:162] - c) ** 2).sum(1)).mean(dtype=np.float32)
        dv = sv - c
        ln = np.sqrt((dv * dv).sum(1, dtype=np.float32)).astype(np.float32)
        sv = (c + dv * (r / ln)[:, None]).astype(np.float32)
    text = obj_text(sv, sf)
    for i, metallic in enumerate((0.0, 0.25, 0.5, 0.75, 1.0)):
        for j, rough in enumerate((0.1, 0.2, 0.3, 0.5, 0.8)):
            m = S.Material.disney(baseColor=(0.75, 0.7, 0.15), metallic=metallic, roughness=rough, clearcoat=1.0)
            hs.readObjText(text, m, S.getTransformMatrix((0, 0, 0), (-2.4 + 1.2 * i, -0.9 + 0.0 * j, -2.4 + 1.2 * j),
                                                         (1.0, 1.0, 1.0)), True)
    qv, qf = mesh("quad")
    hs.readObjText(obj_text(qv, qf), S.Material.disney(baseColor=(0.725, 0.71, 0.68)),
                   S.getTransformMatrix((0, 0, 0), (0, -1.4, 0), (18.83, 0.01, 18.83)), False)
    h = env_map(hdr)
    return _finish("disney_grid_sub%d" % subdiv, hs, leaf_n, h, False, FILTER_NEAREST)
