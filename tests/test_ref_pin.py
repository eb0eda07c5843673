"""Pins "oracle == reference": the repo's host code and the oracle's intersector against
oracle/_ref = the REFERENCE's own C++ compiled from /root/reference by oracle/ref_recipe/build_ref.py
(P2/P3/P4/P5 main.cpp + lib/hdrloader.cpp, unmodified, headless GL/GLM stand-ins).

Tests that feed synthetic inputs run wherever the prebuilt oracle/_ref/*.so exist; tests that open
the reference's asset files (OBJ, HDR) or run a chapter's main() need /root/reference and skip
elsewhere (e.g. on the GPU box).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref as R  # noqa: E402

from ezrt_amd import scene as S  # noqa: E402
from ezrt_amd import scenes  # noqa: E402

pytestmark = pytest.mark.skipif(not all(R.available(t) for t in ("p2", "p3", "p4", "p5")),
                                reason="oracle/_ref not built (python oracle/ref_recipe/build_ref.py)")
needs_assets = pytest.mark.skipif(not R.have_reference(), reason="reference assets not mounted")

P3_MODELS = os.path.join(R.source_dir("p3"), "models")
P4_HDR = os.path.join(R.source_dir("p4"), "HDR", "peppermint_powerplant_4k.hdr")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def _random_tris(n, seed, spread=2.0, size=0.3):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3))
    P = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    T = np.zeros((n, 36), np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 1, 0], 3)
    T[:, 18:36] = S.Material.disney().to18()
    return T


def _rays(n, seed, lo=-3, hi=3):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3))
    tgt = rng.uniform(-1, 1, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# a1: Material defaults of the chapters
def test_material_defaults_are_the_reference_structs():
    assert same_bits(R.Flat("p3").materialDefault(), S.Material().to18())             # P3/main.cpp:28-43
    assert same_bits(R.Flat("p4").materialDefault(), S.Material.disney().to18())      # P4/main.cpp:27-43
    assert same_bits(R.Flat("p5").materialDefault(), S.Material.disney().to18())      # P5/main.cpp:27-42


# a6: getTransformMatrix.  The reference calls glm::rotate -> std::sin/std::cos of the build's libm
# (implementation-defined); the repo defines them through include/ezrt_detmath.h.  Every canned scene
# uses rotate = (0,0,0), where both are exact: bit-equal there, a few ulp elsewhere.
def test_get_transform_matrix_equals_the_reference():
    f = R.Flat("p3")
    rng = np.random.default_rng(1)
    cases = [((0, 0, 0), (0.3, -1.6, 0), (1.5, 1.5, 1.5)), ((0, 0, 0), (0, -1.4, 0), (18.83, 0.01, 18.83)),
             ((0, 0, 0), (0.0, 0.9, -0.0), (1, 1, 1)), ((0, 0, 0), (0, -0.5, 0), (13000.0, 0.01, 13000.0))]
    cases += [((0, 0, 0), rng.uniform(-3, 3, 3), rng.uniform(0.1, 4, 3)) for _ in range(100)]
    for r, t, s in cases:
        assert same_bits(f.getTransformMatrix(r, t, s), S.getTransformMatrix(r, t, s))
    for _ in range(200):
        r, t, s = rng.uniform(-180, 180, 3), rng.uniform(-3, 3, 3), rng.uniform(0.1, 4, 3)
        a, b = f.getTransformMatrix(r, t, s), S.getTransformMatrix(r, t, s)
        assert np.abs(a - b).max() <= 4e-6 * max(1.0, float(np.abs(a).max()))


# a11: display()'s camera (lookAt + inverse in GLM's operand order; sin/cos as above)
def test_camera_equals_the_reference_display():
    f = R.Flat("p5")
    rng = np.random.default_rng(2)
    for rot, up, rad in [(0, 0, 4), (90, 10, 2), (0, 15, 8), (0, 0, 10), (0, 0, 2), (180, 0, 4)]:   # the presets
        e_r, m_r = f.camera(rot, up, rad)
        e_o, m_o = S.camera(rot, up, rad)
        assert same_bits(e_r, e_o) and same_bits(m_r, m_o), (rot, up, rad)
    for rot, up, rad in zip(rng.uniform(-360, 360, 200), rng.uniform(-89, 89, 200), rng.uniform(0.5, 12, 200)):
        e_r, m_r = f.camera(float(rot), float(up), float(rad))
        e_o, m_o = S.camera(float(rot), float(up), float(rad))
        assert np.abs(e_r - e_o).max() <= 4e-6 * rad and np.abs(m_r - m_o).max() <= 4e-6 * max(1.0, rad)


# ----------------------------------------------------------------------------------------------
# a5: readObj on every shipped model, smooth and flat normals, incl. the extent bug
@needs_assets
@pytest.mark.parametrize("fn", ["Stanford Bunny.obj", "quad.obj", "sphere.obj", "sphere2.obj"])
@pytest.mark.parametrize("smooth", [True, False])
def test_read_obj_equals_the_reference(fn, smooth):
    f = R.Flat("p3")
    f.clear()
    mat = S.Material.disney(baseColor=(0.2, 0.5, 0.9), roughness=0.3)
    trans = S.getTransformMatrix((10, 20, 30), (0.3, -1.6, 0.1), (1.5, 0.7, 1.1))
    path = os.path.join(P3_MODELS, fn)
    f.readObj(path, mat.to18(), trans, smooth)
    tri_ref, _, _ = f.scene()
    hs = S.HostScene()
    hs.readObj(path, mat, trans, smooth)
    tri, _ = hs.encode()
    assert tri.shape[0] > 0 and same_bits(tri, tri_ref)


# a7 + a8: the flat builders on the reference scene and on random soups (no equal sort keys)
@needs_assets
@pytest.mark.parametrize("sah,leaf", [(True, 8), (False, 8), (True, 2), (False, 1)])
def test_flat_builders_equal_the_reference_on_the_p3_scene(sah, leaf):
    f = R.Flat("p3")
    f.clear()
    hs = S.HostScene()
    for fn, mat, t, s, smooth in (("Stanford Bunny.obj", S.Material(), (0.3, -1.6, 0), (1.5, 1.5, 1.5), True),
                                  ("quad.obj", S.Material(baseColor=(0.725, 0.71, 0.68)), (0, -1.4, 0), (18.83, 0.01, 18.83), False),
                                  ("sphere.obj", S.Material(emissive=(30, 20, 10)), (0, 0.9, 0), (1, 1, 1), False)):
        trans = S.getTransformMatrix((0, 0, 0), t, s)
        f.readObj(os.path.join(P3_MODELS, fn), mat.to18(), trans, smooth)
        hs.readObj(os.path.join(P3_MODELS, fn), mat, trans, smooth)
    f.build(sah, leaf)
    # the floor box has triangles with exactly equal centroid coordinates: std::sort's order of equal
    # keys is the library's; TieOrder::LibrarySort reproduces it (include/ezrt_scene.hpp)
    S.setTieOrder(True)
    try:
        (hs.buildBVHwithSAH if sah else hs.buildBVH)(leaf)
    finally:
        S.setTieOrder(False)
    tri, nodes = hs.encode()
    tri_ref, _, _ = f.scene()
    assert same_bits(tri, tri_ref)
    assert same_bits(nodes, f.encodedNodes())


def test_equal_keys_follow_the_library_sort_when_asked():
    """every triangle twice (all keys tie pairwise) + a grid with many equal coordinates"""
    T = _random_tris(700, 21)
    g = np.round(T[:, :9] * 2) / 2          # snap to a coarse grid: lots of exactly equal centroids
    T2 = T.copy()
    T2[:, :9] = g + np.tile(np.array([0, 0, 0, 0.25, 0, 0, 0, 0.25, 0], np.float32), (700, 1))
    T = np.concatenate([T, T, T2], 0)
    for sah in (True, False):
        f = R.Flat("p4")
        f.clear()
        f.addTriangles(T)
        f.build(sah, 4)
        hs = S.HostScene()
        hs.addTriangles(T)
        S.setTieOrder(True)
        try:
            (hs.buildBVHwithSAH if sah else hs.buildBVH)(4)
        finally:
            S.setTieOrder(False)
        tri, nodes = hs.encode()
        assert same_bits(tri, f.scene()[0]) and same_bits(nodes, f.encodedNodes())
        hs2 = S.HostScene()
        hs2.addTriangles(T)
        (hs2.buildBVHwithSAH if sah else hs2.buildBVH)(4)
        tri_s, nodes_s = hs2.encode()
        assert nodes_s.shape[0] > 3
        # same multiset of triangles either way
        assert np.array_equal(np.sort(bits(tri_s).view(np.uint32), axis=0), np.sort(bits(tri), axis=0))


@pytest.mark.parametrize("n,sah,leaf,seed", [(1, True, 8, 0), (2, False, 1, 1), (9, True, 8, 2), (500, True, 8, 3),
                                              (500, False, 8, 4), (4000, True, 4, 5), (4000, False, 3, 6),
                                              (20000, True, 8, 7)])
def test_flat_builders_equal_the_reference_on_random_soups(n, sah, leaf, seed):
    T = _random_tris(n, seed)
    f = R.Flat("p5")
    f.clear()
    f.addTriangles(T)
    f.build(sah, leaf)
    hs = S.HostScene()
    hs.addTriangles(T)
    (hs.buildBVHwithSAH if sah else hs.buildBVH)(leaf)
    tri, nodes = hs.encode()
    tri_ref, _, _ = f.scene()
    assert same_bits(tri, tri_ref)
    assert same_bits(nodes, f.encodedNodes())


def test_sah_inf_cap_matches_the_reference():
    """INF = 114514 caps the SAH cost (P3/main.cpp:499, 21): with huge boxes every split costs
    more than INF and the reference falls back to Split = (l + r) / 2 on axis 0."""
    T = _random_tris(300, 9, spread=400.0, size=5.0)
    f = R.Flat("p3")
    f.clear()
    f.addTriangles(T)
    f.build(True, 8)
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVHwithSAH(8)
    assert hs.buildStats()["inf_cap_nodes"] > 0
    tri, nodes = hs.encode()
    assert same_bits(tri, f.scene()[0]) and same_bits(nodes, f.encodedNodes())


# a4 + the whole pre-frame sequence: chapter 3's main(), run headless
@needs_assets
def test_chapter3_main_uploads_exactly_our_arrays():
    """P3/main.cpp:676-816 executed as it is (GL calls recorded, no context): the two
    glBufferData(GL_TEXTURE_BUFFER) payloads are the encoded triangle and node arrays."""
    f = R.Flat("p3")
    bufs = f.runMain()
    assert len(bufs) == 2
    tri_ref = bufs[0].reshape(-1, 36)
    nodes_ref = bufs[1].reshape(-1, 12)
    assert f.uniformInt("nTriangles") == tri_ref.shape[0] == 5300
    assert f.uniformInt("nNodes") == nodes_ref.shape[0] == 1868
    assert (f.uniformInt("width"), f.uniformInt("height")) == (512, 512)
    ours = scenes.bunny_scene(subdiv=0, materials="p3", hdr=None)
    assert same_bits(ours.tri, tri_ref)
    assert same_bits(ours.nodes, nodes_ref)
    # the canned C2 scene differs from it only by the P4/P5 material defaults
    c2 = scenes.bunny_scene(subdiv=0, hdr=None)
    diff = np.nonzero((bits(c2.tri) != bits(tri_ref)).any(0))[0]
    assert diff.tolist() == [26, 28, 31, 33] and same_bits(c2.nodes, nodes_ref)


# ----------------------------------------------------------------------------------------------
# a16/a17/a18: the C++ twins of the intersector (chapter 2) against the trace oracle
def test_hit_triangle_equals_the_reference_cpp_twin(oracle):
    """P2/main.cpp:212-238 == the oracle's hitTriangle (P5/fsh:160-198): same decision, same t."""
    p2 = R.P2()
    rng = np.random.default_rng(3)
    n = 200000
    T = _random_tris(n, 4, spread=1.0, size=0.8)[:, :9]
    rays = _rays(n, 5, -2, 2)
    # a third of the rays aimed at their triangle so that hits are common
    aim = rng.random(n) < 0.6
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    tgt = (T.reshape(n, 3, 3) * w[:, :, None]).sum(1)
    d = tgt - rays[:, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[aim, 3:] = d[aim].astype(np.float32)
    t_ref = p2.hitTriangle(T, rays)
    t_ora = oracle.debug_math(11, rays, T, n)
    assert (t_ref < p2.INF).mean() > 0.3
    assert np.array_equal(bits(t_ref), bits(t_ora))


def test_hit_aabb_equals_the_reference_cpp_twin(oracle):
    """P2/main.cpp:449-463 == the oracle's hitAABB (P5/fsh:220-233), including rays that start
    inside the box, axis-parallel rays (1/0 = inf) and boxes behind the origin."""
    p2 = R.P2()
    rng = np.random.default_rng(6)
    n = 300000
    rays = _rays(n, 7)
    ax = rng.random(n) < 0.15
    k = rng.integers(0, 3, n)
    rays[ax, 3 + k[ax]] = 0.0
    c = rng.uniform(-2, 2, (n, 3))
    e = rng.uniform(0.01, 1.5, (n, 3))
    boxes = np.concatenate([c - e, c + e], 1).astype(np.float32)
    t_ref = p2.hitAABB(rays, boxes)
    t_ora = oracle.debug_math(10, rays, boxes, n)
    assert (t_ref > 0).mean() > 0.2 and (t_ref == -1).mean() > 0.2
    both_nan = np.isnan(t_ref) & np.isnan(t_ora)
    assert np.array_equal(bits(t_ref)[~both_nan], bits(t_ora)[~both_nan])


def test_brute_force_hits_equal_the_reference_on_the_bunny(oracle, bunny_small):
    """hitTriangleArray over the whole scene (P2/main.cpp:436-446; the acceptance check of
    P2/main.cpp:585) == the oracle's hitBVH: same winner distance bit for bit, same winner."""
    p2 = R.P2()
    p2.setTriangles(bunny_small.tri[:, :9])
    rays = _rays(20000, 8)
    i_ref, t_ref = p2.hitTriangleArray(rays)
    so = bunny_small.upload(oracle)
    i_ora, t_ora = so.query_hits(rays)
    hit = i_ref >= 0
    assert hit.sum() > 4000 and np.array_equal(hit, i_ora >= 0)
    assert np.array_equal(bits(t_ref[hit]), bits(t_ora[hit]))
    assert np.all(t_ref[~hit] == np.float32(114514.0))
    assert np.array_equal(i_ref, i_ora)      # the array order is the BVH order, first minimum wins in both


@pytest.mark.parametrize("sah", [True, False])
def test_pointer_tree_builders_equal_the_reference(sah):
    """chapter 2's builders (P2/main.cpp:242-433, precomputed `center`) vs ezrt::p2"""
    T = _random_tris(3000, 10)[:, :9]
    p2 = R.P2()
    p2.setTriangles(T)
    ints, boxes = p2.build(sah, 8)
    rays = _rays(64, 11)
    tris, ib, tb = S.p2Query(T, rays, sah=sah, use_bvh=True)
    assert same_bits(tris, p2.triangles())
    leaves = ints[:, 2] > 0
    assert ints[leaves, 2].sum() == 3000


@needs_assets
def test_chapter2_main_probe_ray():
    """P2/main.cpp:540-588 executed as it is.  Its hitBVH passes the leaf range as
    (root->n, root->n + root->index - 1) -- n and index swapped (P2/main.cpp:471) -- so the probe
    only sees what that range happens to cover; the brute-force line the tutorial keeps commented
    out (585) is the correct answer, and it is what the oracle and ezrt::p2 return."""
    p2 = R.P2()
    lines = p2.runMain()
    tris = p2.triangles()
    assert tris.shape[0] == 4968 + 2       # P2's quad.obj is a two-triangle quad
    ray = np.array([[0, 0, 1, 0.1, -0.1, -0.7]], np.float64)
    ray[0, 3:] /= np.linalg.norm(ray[0, 3:])
    ro = np.zeros((1, 6), np.float32)
    ro[0, :3] = (0, 0, 1)
    # the reference normalises in fp32 (glm::normalize): reproduce through the camera-free path
    v = np.array([0.1, -0.1, -0.7], np.float32)
    inv = np.float32(1.0) / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=np.float32)
    ro[0, 3:] = v * inv
    i_brute, t_brute = p2.hitTriangleArray(ro)
    _, i_ours, t_ours = S.p2Query(tris, ro, sah=True, use_bvh=False)
    assert i_brute[0] >= 0 and np.array_equal(bits(t_brute), bits(t_ours))
    assert float(t_brute[0]) == PROBE_T
    # main()'s own (buggy-range) answer: the triangle it drew is the last addTriangle block of `lines`
    i_bvh, t_bvh = p2.hitBVH(ro)
    if i_bvh[0] >= 0:
        drawn = lines[-14] + np.float32(0.0005)
        assert np.allclose(drawn, tris[i_bvh[0], :3], atol=1e-6)
        assert t_bvh[0] >= t_brute[0]


PROBE_T = 0.7623799443244934  # same value as tests/test_oracle.py::PROBE_T, now produced by the reference's code


# ----------------------------------------------------------------------------------------------
# a10 + a9: HDRLoader::load and calculateHdrCache on the only shipped HDR, every texel
@needs_assets
def test_hdr_loader_and_cache_equal_the_reference_on_the_shipped_hdr():
    f = R.Flat("p5")
    hdr_ref = f.hdrLoad(P4_HDR)
    hdr = S.hdrLoad(P4_HDR)
    assert hdr_ref.shape == (512, 1024, 3) and same_bits(hdr, hdr_ref)
    cache_ref = f.calculateHdrCache(hdr_ref)
    cache = S.calculateHdrCache(hdr)
    assert same_bits(cache, cache_ref)
    # the shipped asset copy (RGBE texels) decodes to the same floats
    assert same_bits(scenes.shipped_hdr(), hdr_ref)


def test_hdr_cache_equals_the_reference_on_synthetic_maps():
    f = R.Flat("p5")
    rng = np.random.default_rng(12)
    for (h, w) in ((12, 20), (64, 128), (33, 17)):
        hdr = (rng.uniform(0.05, 1.0, (h, w, 3)) ** 3 * 5).astype(np.float32)
        hdr[h // 3, w // 3] = 400.0
        assert same_bits(S.calculateHdrCache(hdr), f.calculateHdrCache(hdr))
    hdr = scenes.synthetic_hdr(256, 128)
    assert same_bits(S.calculateHdrCache(hdr), f.calculateHdrCache(hdr))


def test_hdr_loader_equals_the_reference_on_written_files(tmp_path):
    """RLE and flat scanlines written here, decoded by both loaders."""
    rng = np.random.default_rng(13)
    h, w = 6, 40
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(120, 136, (h, w))
    body = b""
    for y in range(h):
        body += bytes([2, 2, w >> 8, w & 255])
        for ch in range(4):
            row = img[y, :, ch]
            x = 0
            while x < w:
                n = min(100, w - x)
                body += bytes([n]) + row[x:x + n].tobytes()
                x += n
    data = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + body
    p = tmp_path / "rle.hdr"
    p.write_bytes(data)
    a = R.Flat("p4").hdrLoad(str(p))
    b = S.hdrLoad(str(p))
    assert a.shape == (h, w, 3) and same_bits(a, b)
    flat = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 5\n" + img[:3, :5].tobytes()
    p = tmp_path / "flat.hdr"
    p.write_bytes(flat)
    assert same_bits(R.Flat("p4").hdrLoad(str(p)), S.hdrLoad(str(p)))
