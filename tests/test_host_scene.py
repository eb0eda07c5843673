"""Host scene-build API (libezrt_scene.so) against independent numpy restatements and the
invariants the reference's tutorials check by eye."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import np_oracle as NPO  # noqa: E402

from ezrt_amd import scene as S  # noqa: E402
from ezrt_amd import scenes  # noqa: E402

P3_MODELS = "/root/reference/part 3 -- OpenGL Raytracing/source code/models"
P4_HDR = "/root/reference/part 4 -- Disney Principle BRDF/source code/HDR/peppermint_powerplant_4k.hdr"


def _random_tris(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-2, 2, (n, 1, 3))
    P = (c + rng.uniform(-0.3, 0.3, (n, 3, 3))).astype(np.float32)
    T = np.zeros((n, 36), np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 1, 0], 3)
    T[:, 18:36] = S.Material.disney().to18()
    return T


def test_testnode_roundtrip_and_root_conventions(bunny_small):
    nodes = bunny_small.nodes
    # node 0 = the testNode sentinel (P3/main.cpp:707-713), ints stored as floats
    assert nodes[0, :2].tolist() == [255.0, 128.0] and nodes[0, 3] == 30.0
    assert nodes[0, 6:12].tolist() == [1, 1, 0, 0, 1, 0]
    assert nodes[:, [0, 1, 3, 4]].astype(np.int32).astype(np.float32).tolist() == nodes[:, [0, 1, 3, 4]].tolist()
    # pre-order allocation: root is node 1 and its left child is node 2 (T3:601-628)
    assert int(nodes[1, 0]) == 2
    assert (nodes[:, 2] == 0).all() and (nodes[:, 5] == 0).all()


def test_leaves_partition_the_triangle_array(bunny_small):
    nodes = bunny_small.nodes[1:]
    leaf = nodes[:, 3] > 0
    idx = nodes[leaf, 4].astype(int)
    cnt = nodes[leaf, 3].astype(int)
    order = np.argsort(idx)
    idx, cnt = idx[order], cnt[order]
    assert idx[0] == 0 and (idx[1:] == idx[:-1] + cnt[:-1]).all() and idx[-1] + cnt[-1] == bunny_small.tri.shape[0]
    assert cnt.max() <= 8
    inner = nodes[~leaf]
    assert (inner[:, 0] > 0).all() and (inner[:, 1] > 0).all()
    assert bunny_small.tri.shape == (5300, 36) and bunny_small.nodes.shape == (1868, 12)


def test_node_boxes_bound_their_triangles(bunny_small):
    nodes, tri = bunny_small.nodes, bunny_small.tri
    for i in range(1, nodes.shape[0]):
        if nodes[i, 3] > 0:
            a, n = int(nodes[i, 4]), int(nodes[i, 3])
            P = tri[a:a + n, :9].reshape(-1, 3)
            assert (P.min(0) == nodes[i, 6:9]).all() and (P.max(0) == nodes[i, 9:12]).all()
        else:
            for c in (int(nodes[i, 0]), int(nodes[i, 1])):
                assert (nodes[c, 6:9] >= nodes[i, 6:9]).all() and (nodes[c, 9:12] <= nodes[i, 9:12]).all()


@pytest.mark.parametrize("sah", [True, False])
@pytest.mark.parametrize("n,seed", [(9, 1), (40, 2), (257, 3)])
def test_builders_match_numpy_restatement(n, seed, sah):
    T = _random_tris(n, seed)
    hs = S.HostScene()
    hs.addTriangles(T)
    (hs.buildBVHwithSAH if sah else hs.buildBVH)(8)
    tri, nodes = hs.encode()
    order, ref_nodes = NPO.build_bvh(T, 8, sah)
    assert np.array_equal(tri, T[order])
    assert np.array_equal(nodes, NPO.encode_nodes(ref_nodes))


def test_sah_inf_cap_falls_back_to_median_x():
    """Every candidate cost >= INF=114514 keeps Axis=0, Split=(l+r)/2 (P3/main.cpp:492-494)."""
    T = _random_tris(64, 7)
    T[:, :9] *= 400.0  # areas ~ 1e5..1e6 so area*count >= 114514 at the top
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVHwithSAH(8)
    st = hs.buildStats()
    assert st["inf_cap_nodes"] >= 1
    tri, nodes = hs.encode()
    order, ref_nodes = NPO.build_bvh(T, 8, True)
    assert np.array_equal(tri, T[order]) and np.array_equal(nodes, NPO.encode_nodes(ref_nodes))
    left = int(nodes[1, 0])
    # median split of 64 triangles: left subtree holds [0, 31]
    first_right = int(nodes[int(nodes[1, 1]), 4]) if nodes[int(nodes[1, 1]), 3] > 0 else None
    assert left == 2 and first_right in (None, 32)


def test_read_obj_extent_bug_and_normals():
    """maxy/maxz/miny/minz are taken against maxx/minx (P3/main.cpp:316-317): the divisor is
    max over (x extent, last-vertex-dependent y/z terms), not the true max extent."""
    text = b"v 0 0 0\nv 2 0 0\nv 0 8 0\nv 1 1 1\nf 1 2 3\nf 1/1/1 2/2/2 4/4/4\n"
    hs = S.HostScene()
    ident = S.getTransformMatrix((0, 0, 0), (0, 0, 0), (1, 1, 1))
    hs.readObjText(text, S.Material(), ident, False)
    hs.buildBVH(8)
    tri, _ = hs.encode()
    # maxx=2, minx=0; last vertex (1,1,1): maxy=max(2,1)=2, miny=min(0,1)=0 -> divisor 2, not 8
    P = tri[:, :9].reshape(-1, 3)
    assert P.max() == 4.0  # 8 / 2
    n = tri[0, 9:12]
    assert np.allclose(n, [0, 0, 1])


def test_fixture_obj_text_roundtrips_float32():
    v, f = scenes.mesh("bunny")
    assert v.shape == (2503, 3) and f.shape == (4968, 3)
    txt = scenes.obj_text(v[:50], f[:0]).decode().split("\n")
    back = np.array([[np.float32(x) for x in ln.split()[1:]] for ln in txt if ln.startswith("v ")], np.float32)
    assert np.array_equal(back, v[:50])


@pytest.mark.skipif(not os.path.isdir(P3_MODELS), reason="reference not mounted")
def test_fixture_scene_equals_reference_obj_files():
    """Scene built from the committed mesh fixture == scene built from the reference's OBJ files."""
    a = scenes.bunny_scene(subdiv=0, hdr=None)
    hs = S.HostScene()
    mk = S.Material.disney
    hs.readObj(os.path.join(P3_MODELS, "Stanford Bunny.obj"), mk(baseColor=(1, 1, 1)),
               S.getTransformMatrix((0, 0, 0), (0.3, -1.6, 0), (1.5, 1.5, 1.5)), True)
    hs.readObj(os.path.join(P3_MODELS, "quad.obj"), mk(baseColor=(0.725, 0.71, 0.68)),
               S.getTransformMatrix((0, 0, 0), (0, -1.4, 0), (18.83, 0.01, 18.83)), False)
    hs.readObj(os.path.join(P3_MODELS, "sphere.obj"), mk(baseColor=(1, 1, 1), emissive=(30, 20, 10)),
               S.getTransformMatrix((0, 0, 0), (0.0, 0.9, -0.0), (1, 1, 1)), False)
    hs.buildBVHwithSAH(8)
    tri, nodes = hs.encode()
    assert np.array_equal(tri, a.tri) and np.array_equal(nodes, a.nodes)


def test_read_obj_missing_file_raises_instead_of_exit():
    hs = S.HostScene()
    with pytest.raises(RuntimeError, match="cannot open"):
        hs.readObj("/nonexistent/x.obj", S.Material(), np.eye(4, dtype=np.float32).ravel(), False)


def test_transform_matrix_and_camera():
    m = S.getTransformMatrix((0, 0, 0), (1, 2, 3), (2, 2, 2)).reshape(4, 4)  # columns
    assert np.array_equal(m[3], [1, 2, 3, 1]) and np.array_equal(np.diag(m), [2, 2, 2, 1])
    r = S.getTransformMatrix((0, 90, 0), (0, 0, 0), (1, 1, 1)).reshape(4, 4)
    assert np.allclose(r[0, :3], [0, 0, -1], atol=1e-6) and np.allclose(r[2, :3], [1, 0, 0], atol=1e-6)
    eye, cam = S.camera(0, 0, 4)
    assert np.allclose(eye, [0, 0, 4], atol=1e-6)
    c = cam.reshape(4, 4)
    assert np.allclose(c[:3, :3], np.eye(3), atol=1e-6) and np.allclose(c[3, :3], [0, 0, 4], atol=1e-5)
    eye, cam = S.camera(90, 10, 2)  # P5 preset
    assert np.allclose(np.linalg.norm(eye), 2, atol=1e-5) and eye[0] < -1.9
    c = cam.reshape(4, 4)[:3, :3]
    assert np.allclose(c @ c.T, np.eye(3), atol=1e-5)


def test_material_defaults():
    assert S.Material().to18().tolist() == [0, 0, 0, 1, 1, 1] + [0] * 10 + [1, 0]
    d = S.Material.disney().to18()
    assert d[8] == 0.5 and d[10] == 0.5 and d[13] == 0.5 and d[15] == 1.0


# ----------------------------------------------------------------------------- HDR
def _rle_hdr(img_rgbe):
    """Encode uint8 [h, w, 4] RGBE as a new-style RLE Radiance file (runs + literals)."""
    h, w, _ = img_rgbe.shape
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode())
    for y in range(h):
        out += bytes([2, 2, (w >> 8) & 255, w & 255])
        for c in range(4):
            row = img_rgbe[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    n = min(w - x, 5)
                    out += bytes([n]) + bytes(int(v) for v in row[x:x + n])
                    x += n
    return bytes(out)


def test_hdr_loader_decodes_rle_and_flat_files():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (6, 16, 4), dtype=np.uint8)
    img[:, 4:12, 0] = 77  # force runs
    img[..., 3] = rng.integers(120, 136, (6, 16))
    want = (img[..., :3].astype(np.float32) / np.float32(256.0)) * np.exp2(img[..., 3:4].astype(np.float32) - 128)
    got = S.hdrLoad(data=_rle_hdr(img))
    assert got.shape == (6, 16, 3) and np.array_equal(got, want.astype(np.float32))
    flat = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 3\n" + img[:2, :3].tobytes()  # width < 8: old flat path
    got = S.hdrLoad(data=flat)
    assert np.array_equal(got, want[:2, :3].astype(np.float32))
    with pytest.raises(RuntimeError):
        S.hdrLoad(data=b"not a radiance file at all")


def _bits_sum(a):
    return int(np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64).sum())


def test_shipped_hdr_asset_and_its_cache_match_the_reference_produced_probes():
    """tests/golden/hdr_probe.json holds texels + checksums of P4/HDR/peppermint_powerplant_4k.hdr as
    decoded by the REFERENCE's HDRLoader::load and of the REFERENCE's calculateHdrCache on it (made by
    tests/golden/make_fixtures.py through oracle/_ref).  The asset copy and our cache must reproduce them."""
    import json
    info = json.load(open(os.path.join(ROOT, "tests", "golden", "hdr_probe.json")))
    assert "oracle/_ref" in info["produced_by"]
    hdr = scenes.shipped_hdr()
    assert hdr.shape == (info["height"], info["width"], 3) == (512, 1024, 3)
    cache = S.calculateHdrCache(hdr)
    for pr in info["probes"]:
        assert [int(x) for x in hdr[pr["row"], pr["col"]].view(np.uint32)] == pr["rgb_bits"]
        assert [int(x) for x in cache[pr["row"], pr["col"]].view(np.uint32)] == pr["cache_bits"]
    assert int(np.bitwise_xor.reduce(hdr.view(np.uint32).ravel())) == info["xor_bits"]
    assert _bits_sum(hdr) == info["sum_bits"]
    assert int(np.bitwise_xor.reduce(cache.view(np.uint32).ravel())) == info["cache_xor_bits"]
    assert _bits_sum(cache) == info["cache_sum_bits"]
    if os.path.exists(P4_HDR):                      # our loader on the file itself
        assert np.array_equal(S.hdrLoad(P4_HDR).view(np.uint32), hdr.view(np.uint32))


def test_canned_p3_scene_matches_the_reference_main_checksums():
    """tests/golden/ref_scene_p3.json = what chapter 3's main() uploads (run headless through
    oracle/_ref): the canned scene rebuilt from the mesh assets gives the same two arrays."""
    import json
    info = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_scene_p3.json")))
    b = scenes.bunny_scene(subdiv=0, materials="p3", hdr=None)
    assert b.tri.shape[0] == info["nTriangles"] and b.nodes.shape[0] == info["nNodes"]
    assert int(np.bitwise_xor.reduce(b.tri.view(np.uint32).ravel())) == info["tri_xor_bits"]
    assert _bits_sum(b.tri) == info["tri_sum_bits"]
    assert int(np.bitwise_xor.reduce(b.nodes.view(np.uint32).ravel())) == info["nodes_xor_bits"]
    assert _bits_sum(b.nodes) == info["nodes_sum_bits"]
    assert b.nodes[1].tolist() == info["node1"] and b.tri[0].tolist() == info["tri0"]


def test_hdr_cache_matches_numpy_restatement():
    rng = np.random.default_rng(11)
    hdr = (rng.uniform(0.05, 1.0, (12, 20, 3)) ** 3 * 5).astype(np.float32)
    hdr[3, 7] = 400.0  # a "sun"
    got = S.calculateHdrCache(hdr)
    want = NPO.hdr_cache(hdr)
    assert np.array_equal(got, want)
    assert abs(float(got[..., 2].astype(np.float64).sum()) - 1.0) < 1e-4
    # importance: the hot texel's column is what most xi_1 rows map to
    assert (got[..., 0] == np.float32(7) / np.float32(20)).mean() > 0.5


def test_synthetic_hdr_is_deterministic():
    a = scenes.synthetic_hdr(64, 32)
    assert a.dtype == np.float32 and a.shape == (32, 64, 3) and np.isfinite(a).all() and a.min() > 0
    full = scenes.synthetic_hdr()
    assert full.shape == (512, 1024, 3) and full.max() > 50
    assert struct.unpack("<I", np.bitwise_xor.reduce(full.view(np.uint32).ravel()).tobytes())[0] == FULL_HDR_XOR


FULL_HDR_XOR = 8264391


def _probe_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3, 3, (n, 3))
    tgt = rng.uniform(-1, 1, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("sah", [True, False])
def test_p2_pointer_tree_query_equals_brute_force_and_the_oracle(oracle, bunny_small, sah):
    """Chapter 2's acceptance check (P2/main.cpp:581-588: hitBVH == hitTriangleArray over the whole
    array) on the host pointer-tree API, and the same distances, bit for bit, as the trace oracle."""
    tri9 = bunny_small.tri[:, :9]
    rays = _probe_rays(3000, 5)
    tris, ib, tb = S.p2Query(tri9, rays, sah=sah, use_bvh=True)
    _, ia, ta = S.p2Query(tri9, rays, sah=sah, use_bvh=False)
    assert np.array_equal(tb.view(np.uint32), ta.view(np.uint32))
    assert (ib >= 0).sum() > 500
    assert np.array_equal(np.sort(tris.view(np.uint32), axis=0), np.sort(tri9.view(np.uint32), axis=0))
    so = bunny_small.upload(oracle)
    to_tri, to_t = so.query_hits(rays)
    hit = to_tri >= 0
    assert np.array_equal(hit, ib >= 0)
    assert np.array_equal(tb[hit].view(np.uint32), to_t[hit].view(np.uint32))
    assert np.all(tb[~hit] == np.float32(114514.0))
    same = np.all(tris[ib[hit]] == tri9[to_tri[hit]], axis=1)      # identical triangle unless an exact tie in t
    assert same.mean() > 0.99


def test_c5_scene_has_exactly_a_million_triangles_and_hits_the_sah_cap():
    c5 = scenes.mega_scene()
    assert c5.tri.shape == (1_000_000, 36)
    st = c5.build_stats
    assert st["inf_cap_nodes"] > 0 and st["max_depth"] <= 150
    n = c5.nodes[1:, 3].astype(np.int64)              # leafInfo.x = n
    assert n[n > 0].sum() == 1_000_000 and n.max() <= 8


def test_png_and_pfm_writers(tmp_path):
    import zlib
    from ezrt_amd import imageio
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 2, (5, 7, 4)).astype(np.float32)
    imageio.write_pfm(tmp_path / "a.pfm", img)
    assert np.array_equal(imageio.read_pfm(tmp_path / "a.pfm"), img[..., :3])
    q = imageio.quantize_p1(img[..., :3])
    assert q.dtype == np.uint8 and q.max() == 255 and q[img[..., :3] < 1].max() < 255
    imageio.write_png(tmp_path / "a.png", q)
    raw = open(tmp_path / "a.png", "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and raw[12:16] == b"IHDR"
    w, h = struct.unpack(">II", raw[16:24])
    assert (w, h) == (7, 5)
    i = raw.index(b"IDAT")
    n = struct.unpack(">I", raw[i - 4:i])[0]
    rows = np.frombuffer(zlib.decompress(raw[i + 4:i + 4 + n]), np.uint8).reshape(5, 1 + 7 * 3)
    assert np.array_equal(rows[:, 1:].reshape(5, 7, 3), q[::-1])     # top row first


def test_scene_description_file_builds_the_canned_scene_on_the_host(bunny_small, tmp_path):
    """ezrt_amd.render.build_scene (host builders only here): the P3 scene as JSON gives the arrays of the
    canned scene; "median" selects buildBVH."""
    from ezrt_amd import render
    for name in ("bunny", "quad", "sphere"):
        v, f = scenes.mesh(name)
        (tmp_path / (name + ".obj")).write_bytes(scenes.obj_text(v, f))
    desc = {"integrator": 51, "env": {"synthetic": True}, "bvh": {"builder": "sah", "leaf": 8}, "objects": [
        {"obj": "bunny.obj", "smooth": True, "translate": [0.3, -1.6, 0], "scale": [1.5, 1.5, 1.5],
         "material": {"defaults": "p4", "baseColor": [1, 1, 1]}},
        {"obj": "quad.obj", "translate": [0, -1.4, 0], "scale": [18.83, 0.01, 18.83],
         "material": {"defaults": "p4", "baseColor": [0.725, 0.71, 0.68]}},
        {"obj": "sphere.obj", "translate": [0.0, 0.9, 0.0],
         "material": {"defaults": "p4", "baseColor": [1, 1, 1], "emissive": [30, 20, 10]}}]}
    b = render.build_scene(desc, str(tmp_path))
    assert np.array_equal(b.tri, bunny_small.tri) and np.array_equal(b.nodes, bunny_small.nodes)
    assert b.cache is not None and np.array_equal(b.cache, bunny_small.cache)
    desc["bvh"] = {"builder": "median", "leaf": 4}
    m = render.build_scene(desc, str(tmp_path))
    n = m.nodes[1:, 3]
    assert n.max() <= 4 and n[n > 0].sum() == b.tri.shape[0] and not np.array_equal(m.nodes[:20], b.nodes[:20])
    assert render.material_from({"defaults": "p3", "roughness": 0.25}).to18()[10] == np.float32(0.25)


def test_old_style_runs_edge_cases():
    """(a) four run markers in a row would shift an int by 24+ bits (undefined in the reference,
    hdrloader.cpp:200): refused; (b) decrunch's fallback enters oldDecrunch at pixel 1 with pixel 0 already written
    (hdrloader.cpp:135-139), so a run marker right there repeats pixel 0 as the reference does."""
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n"
    px = lambda r, g, b, e: bytes([r, g, b, e])
    # (b) first byte 2 but second not 2: flat data whose first pixel is (2, 9, 9, 128); then a run of 3, then 4 literals
    body = px(2, 9, 9, 128) + px(1, 1, 1, 3) + b"".join(px(10 + k, 20, 30, 129) for k in range(4))
    got = S.hdrLoad(data=head + body)
    assert got.shape == (1, 8, 3)
    first = np.array([2, 9, 9], np.float32) / np.float32(256.0)
    assert np.array_equal(got[0, :4], np.tile(first, (4, 1)))
    assert np.array_equal(got[0, 4:, 0], (np.arange(10, 14, dtype=np.float32) / np.float32(256.0)) * np.float32(2.0))
    # a run marker as the very first pixel of a line has nothing to repeat: the line fails, and (like the reference,
    # hdrloader.cpp:88-89 `if (decrunch(..) == false) break;`) load still returns, with the unread rows left zero
    assert not S.hdrLoad(data=head + px(1, 1, 1, 2) + px(5, 5, 5, 128) * 8).any()
    # (a) pixel, then four consecutive run markers (1 copy, 0 << 8, 0 << 16, then the shift by 24)
    bad = px(7, 7, 7, 128) + px(1, 1, 1, 1) + px(1, 1, 1, 0) + px(1, 1, 1, 0) + px(1, 1, 1, 1) + px(5, 5, 5, 128) * 8
    assert not S.hdrLoad(data=head + bad).any()
    # three in a row are still fine: 1 + (1 << 8 capped by the line) copies
    ok = px(7, 7, 7, 128) + px(1, 1, 1, 1) + px(1, 1, 1, 1)
    got = S.hdrLoad(data=head + ok)
    assert np.array_equal(got[0], np.tile(np.float32(7) / np.float32(256.0), (8, 3)))
