"""The CPU oracle, pinned against what the reference pins and against
independent numpy restatements; plus the size-independent properties reused by the GPU tests."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import np_oracle as NPO  # noqa: E402

from ezrt_amd import scene as S  # noqa: E402
from ezrt_amd import scenes, trace  # noqa: E402


def camera_rays(n, seed, r=4.0):
    rng = np.random.default_rng(seed)
    o = np.tile(np.array([0, 0, r], np.float32), (n, 1))
    d = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n), -1.5 * np.ones(n)], 1).astype(np.float32)
    d = d / np.sqrt((d * d).sum(1, keepdims=True)).astype(np.float32)
    return np.concatenate([o, d.astype(np.float32)], 1)


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.sqrt((d * d).sum(1, keepdims=True))).astype(np.float32)
    return np.concatenate([o, d], 1)


def test_wang_hash_known_answers(oracle):
    """rand() = float(wang_hash(seed)) / 2^32 (P5/fsh:320-331) against a Python restatement."""
    def wang(s):
        s = ((s ^ 61) ^ (s >> 16)) & 0xFFFFFFFF
        s = (s * 9) & 0xFFFFFFFF
        s = s ^ (s >> 4)
        s = (s * 0x27d4eb2d) & 0xFFFFFFFF
        return s ^ (s >> 15)
    seeds = np.array([1, 2, 1973 | 1, 0xFFFFFFFF, 26699, 123456789, 0x80000000], np.uint32)
    got = oracle.debug_math(9, seeds.view(np.float32))
    want = np.array([np.float32(wang(int(s))) / np.float32(4294967296.0) for s in seeds], np.float32)
    assert np.array_equal(got, want)
    assert ((got >= 0) & (got <= 1)).all()


def test_detmath_accuracy_against_float64(oracle):
    """include/ezrt_detmath.h are *definitions*; they still have to be good built-ins."""
    x = np.linspace(-6.5, 6.5, 200001, dtype=np.float32)
    assert np.abs(oracle.debug_math(0, x) - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(oracle.debug_math(1, x) - np.cos(x.astype(np.float64))).max() < 1.5e-7
    u = np.linspace(-1, 1, 200001, dtype=np.float32)
    assert np.abs(oracle.debug_math(3, u) - np.arcsin(u.astype(np.float64))).max() < 4e-7
    a = np.linspace(0, 2 * np.pi, 100001)
    yy, xx = (np.sin(a) * 3).astype(np.float32), (np.cos(a) * 3).astype(np.float32)
    assert np.abs(oracle.debug_math(2, yy, xx) - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 5e-7
    p = np.geomspace(1e-7, 100, 100001).astype(np.float32)
    lg = oracle.debug_math(4, p)
    assert (np.abs(lg - np.log(p.astype(np.float64))) <= 2e-7 * np.maximum(1, np.abs(np.log(p.astype(np.float64))))).all()
    e = np.linspace(-20, 10, 100001, dtype=np.float32)
    ex = oracle.debug_math(5, e)
    assert (np.abs(ex / np.exp(e.astype(np.float64)) - 1) < 3e-7).all()
    base = np.float32(1e-6) + np.linspace(0, 0.01, 100001, dtype=np.float32)
    ee = np.linspace(0, 1, 100001, dtype=np.float32)
    pw = oracle.debug_math(6, base, ee)
    assert (np.abs(pw / np.power(base.astype(np.float64), ee.astype(np.float64)) - 1) < 3e-6).all()
    assert oracle.debug_math(6, np.array([0, 2, 0.5], np.float32), np.array([0.45, 0, 1], np.float32)).tolist()[:2] == [0, 1]


def test_bvh_equals_brute_force(oracle, bunny_small):
    """The reference's own commented-out check (P2/main.cpp:585): hitBVH == linear scan."""
    sc = bunny_small.upload(oracle)
    rays = np.concatenate([camera_rays(150, 1), random_rays(150, 2)])
    tri, t = sc.query_hits(rays)
    n_hit = 0
    for k in range(rays.shape[0]):
        bi, bt = NPO.brute_force(bunny_small.tri, rays[k, :3], rays[k, 3:])
        assert t[k] == bt, (k, t[k], bt)
        if bi != tri[k]:  # only legal on an exact distance tie between two triangles
            assert bi >= 0 and tri[k] >= 0
            assert NPO.hit_triangles(bunny_small.tri, rays[k, :3], rays[k, 3:])[tri[k]] == bt
        n_hit += bi >= 0
    assert 60 < n_hit < 300


def test_p2_probe_ray_regression(oracle):
    """P2/main.cpp:555-588: Bunny x5, y-0.5, + the two-triangle quad, SAH leaf 8, probe ray from
    (0,0,1) towards normalize(0.1,-0.1,-0.7).  The reference only *draws* the answer (value
    unpinned); here BVH == brute force and the result is frozen as a regression value."""
    bv, bf = scenes.mesh("bunny")
    v = (bv * np.float32(5.0)).astype(np.float32)
    v[:, 1] -= np.float32(0.5)
    qv = np.array([[-0.9, -0.30, -0.9], [-0.9, -0.35, 0.9], [0.9, -0.35, -0.9], [0.9, -0.30, 0.9]], np.float32)
    qf = np.array([[0, 1, 2], [1, 2, 3]]) + v.shape[0]  # P2/models/quad.obj: f 1 2 3 / f 2 3 4
    V = np.concatenate([v, qv])
    F = np.concatenate([bf, qf])
    T = np.zeros((F.shape[0], 36), np.float32)
    T[:, :9] = V[F].reshape(-1, 9)
    T[:, 18:36] = S.Material().to18()
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVHwithSAH(8)
    tri, nodes = hs.encode()
    sc = oracle.scene_create(tri, nodes)
    d = np.array([0.1, -0.1, -0.7], np.float32)
    d = d * (np.float32(1.0) / np.sqrt(np.float32(d @ d)))
    ray = np.concatenate([[0, 0, 1], d]).astype(np.float32)
    ti, tt = sc.query_hits(ray[None])
    bi, bt = NPO.brute_force(tri, ray[:3], ray[3:])
    assert ti[0] == bi and tt[0] == bt and bi >= 0
    print("probe hit: tri %d t %r" % (ti[0], float(tt[0])))
    assert float(tt[0]) == PROBE_T  # frozen regression value


PROBE_T = 0.7623799443244934


def test_counters_and_ray_accounting(oracle, bunny_small):
    sc = bunny_small.upload(oracle)
    eye, cam = S.camera(0, 0, 4)
    sc.set_instrumentation(1)
    p = trace.make_params(48, 48, eye, cam, 50, 4, spp=2)
    sc.render(p)
    c = sc.counters()
    assert c["samples"] == 48 * 48 * 2
    assert c["samples"] <= c["rays"] <= c["samples"] * 5
    assert c["node_pops"] >= c["rays"] and c["inner_pops"] <= c["node_pops"]
    assert c["tri_tests"] >= c["mat_fetch"] > 0
    assert c["env_cache"] == 0 and c["env_map"] > 0
    # integrator 51 shoots up to 1 + 2*bounces rays and reads the cache
    sc.counters_reset()
    p51 = trace.make_params(32, 32, eye, cam, 51, 2, spp=1)
    sc.render(p51)
    c = sc.counters()
    assert c["samples"] == 1024 and c["rays"] <= 1024 * 5 and c["env_cache"] > 0
    sc.set_instrumentation(0)
    sc.counters_reset()
    sc.render(p51)
    c0 = sc.counters()
    assert c0["rays"] == c["rays"] and c0["node_pops"] == 0


@pytest.mark.parametrize("integ,bounces", [(3, 2), (4, 4), (50, 4), (51, 2)])
def test_render_properties(oracle, bunny_small, integ, bounces):
    """Size-independent properties: determinism, frame-range splitting (the running mean is a
    recurrence), pixel-rect and tile-shard decompositions reproduce the full image bit for bit."""
    sc = bunny_small.upload(oracle)
    eye, cam = S.camera(20, 10, 3)
    W, H = 40, 24
    kw = dict(env_clamp=10.0) if integ == 3 else {}
    full = sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=4, **kw))
    assert np.isfinite(full).all() and full[..., :3].max() > 0 and (full[..., 3] == 1).all()
    again = sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=4, **kw))
    assert np.array_equal(full, again)
    # frames 0..1 then 2..3 == frames 0..3
    part = sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=2, frame0=0, **kw))
    part = sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=2, frame0=2, **kw), part)
    assert np.array_equal(full, part)
    # two rects
    img = np.zeros((H, W, 4), np.float32)
    sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=4, rect=(0, 0, 17, H), **kw), img)
    sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=4, rect=(17, 0, W, H), **kw), img)
    assert np.array_equal(full, img)
    # 3 virtual ranks, 8x8 tiles, round-robin
    img = np.zeros((H, W, 4), np.float32)
    for r in range(3):
        sc.render(trace.make_params(W, H, eye, cam, integ, bounces, spp=4, tile=(8, 8), shard=(r, 3), **kw), img)
    assert np.array_equal(full, img)


def test_paths_log_consistent_with_render(oracle, bunny_small):
    sc = bunny_small.upload(oracle)
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(32, 32, eye, cam, 51, 2, spp=1, frame0=3)
    tri, t, col = sc.render_paths(p)
    assert tri.shape == (32, 32, 5)
    assert ((tri[..., 0] >= -1)).all()            # primary always shot
    miss = tri[..., 0] == -1
    assert (tri[miss][:, 1:] == -2).all()          # no bounce after a primary miss
    assert (t[tri < 0] == np.float32(114514.0)).all() and (t[tri >= 0] < 114514.0).all()
    # the sample colour is what render() accumulates for that frame
    acc = np.zeros((32, 32, 4), np.float32)
    one = sc.render(trace.make_params(32, 32, eye, cam, 51, 2, spp=1, frame0=0), acc.copy())
    tri0, _, col0 = sc.render_paths(trace.make_params(32, 32, eye, cam, 51, 2, spp=1, frame0=0))
    assert np.array_equal(one[..., :3], col0)


def test_invalid_arguments_are_errors_not_crashes(oracle, bunny_small):
    with pytest.raises(trace.TraceError):
        oracle.scene_create(bunny_small.tri, bunny_small.nodes[:1])
    bad = bunny_small.nodes.copy()
    bad[1, 0] = 0  # inner node without a left child
    with pytest.raises(trace.TraceError, match="children"):
        oracle.scene_create(bunny_small.tri, bad)
    bad = bunny_small.nodes.copy()
    leaf = np.where(bad[:, 3] > 0)[0][-1]
    bad[leaf, 4] = 10 ** 6
    with pytest.raises(trace.TraceError, match="range"):
        oracle.scene_create(bunny_small.tri, bad)
    sc = oracle.scene_create(bunny_small.tri, bunny_small.nodes)
    eye, cam = S.camera()
    with pytest.raises(trace.TraceError, match="cache"):
        sc.render(trace.make_params(8, 8, eye, cam, 51, 2))
    with pytest.raises(trace.TraceError, match="integrator"):
        sc.render(trace.make_params(8, 8, eye, cam, 7, 2))
    with pytest.raises(trace.TraceError, match="rect"):
        sc.render(trace.make_params(8, 8, eye, cam, 50, 2, rect=(0, 0, 9, 8)))


def test_tonemap_matches_formula(oracle):
    rng = np.random.default_rng(3)
    rgba = rng.uniform(0, 6, (500, 4)).astype(np.float32)
    rgba[0] = 0
    out = oracle.tonemap(rgba)
    c = rgba[:, :3].astype(np.float64)
    lum = 0.3 * c[:, 0] + 0.6 * c[:, 1] + 0.1 * c[:, 2]
    want = np.clip((c / (1 + lum / 1.5)[:, None]) ** (1 / 2.2) * 255, 0, 255)
    assert np.abs(out.astype(np.float64) - np.floor(want)).max() <= 1
    assert out[0].tolist() == [0, 0, 0]
