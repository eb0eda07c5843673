"""Distance pruning of traceq4_kernel (ezrt_amd/csrc/hip/ezrt_traceq4.h "Distance pruning") is PROVEN results-neutral;
these tests attack the proof's corners through the TIMED kernels (audit_via_queue): grazing rays (|N.d| down to the
intersector's 1e-5 threshold), slivers and needles (the 1/sin(theta/2) term), triangles lying in box faces, rays through
vertices and along edges, axis-parallel rays, origins far outside the scene (the |S| term), exact duplicates (ties must
still reach the redo list), the stack-overflow route of the nearest-first order, and scenes the bound does not cover
(leaf boxes that do not hold their triangles: pruning must switch itself off).  Everything is compared with the oracle --
which never prunes -- on the bits, and the three modes with each other."""
import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _build(T, leaf=4):
    hs = S.HostScene()
    hs.addTriangles(np.ascontiguousarray(T, np.float32))
    hs.buildBVHwithSAH(leaf)
    return hs.encode()


def _tri_array(P):
    n = P.shape[0]
    T = np.zeros((n, 36), np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 0, 1], 3)
    T[:, 18:36] = S.Material.disney(baseColor=(0.8, 0.6, 0.4)).to18()
    return T


def _nasty_triangles(rng):
    parts = []
    # ordinary soup
    c = rng.uniform(-2, 2, (1500, 1, 3))
    parts.append(c + rng.uniform(-0.15, 0.15, (1500, 3, 3)))
    # slivers: one long edge, the third vertex a hair off it (min angle 1e-5 .. 1e-2 rad)
    p1 = rng.uniform(-2, 2, (600, 3))
    e = rng.normal(size=(600, 3)); e /= np.linalg.norm(e, axis=1, keepdims=True)
    o = np.cross(e, rng.normal(size=(600, 3))); o /= np.linalg.norm(o, axis=1, keepdims=True)
    w = 10.0 ** rng.uniform(-5, -2, (600, 1))
    parts.append(np.stack([p1, p1 + 2.0 * e, p1 + rng.uniform(0.2, 1.8, (600, 1)) * e + w * o], 1))
    # a tiled plane z = 0.25 (rays can graze it) and axis-aligned quads whose triangles lie IN their boxes' faces
    g = np.linspace(-2, 2, 17)
    for i in range(16):
        for j in range(16):
            a, b = np.array([g[i], g[j], 0.25]), np.array([g[i + 1], g[j + 1], 0.25])
            parts.append(np.array([[[a[0], a[1], .25], [b[0], a[1], .25], [b[0], b[1], .25]], [[a[0], a[1], .25], [b[0], b[1], .25], [a[0], b[1], .25]]]))
    # needles far from the origin (large coordinates: the m_T term) and exact duplicates of 100 triangles (ties)
    far = rng.uniform(-1, 1, (100, 1, 3)) + np.array([40.0, -35.0, 30.0]) + rng.uniform(-0.5, 0.5, (100, 3, 3)) * np.array([1.0, 1e-3, 1.0])
    parts.append(far)
    P = np.concatenate(parts).astype(np.float32)
    P = np.concatenate([P, P[:100]])
    return P


def _nasty_rays(P, rng, n_each=20000):
    rays = []
    n = P.shape[0]
    # random
    o = rng.uniform(-3, 3, (n_each, 3)); t = rng.uniform(-2, 2, (n_each, 3)); d = t - o
    rays.append(np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1))
    # grazing: through a point of a triangle, direction in its plane + a tiny normal component
    k = rng.integers(0, n, n_each)
    bc = rng.dirichlet([1, 1, 1], n_each)
    pt = (P[k] * bc[:, :, None]).sum(1)
    e1 = P[k, 1] - P[k, 0]; e2 = P[k, 2] - P[k, 0]
    N = np.cross(e1, e2); N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-30)
    u = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-30)
    v = np.cross(N, u)
    ang = rng.uniform(0, 2 * np.pi, (n_each, 1))
    inplane = np.cos(ang) * u + np.sin(ang) * v
    tilt = 10.0 ** rng.uniform(-6, -2, (n_each, 1)) * rng.choice([-1, 1], (n_each, 1))
    d = inplane + tilt * N; d /= np.linalg.norm(d, axis=1, keepdims=True)
    L = rng.uniform(0.5, 4.0, (n_each, 1))
    rays.append(np.concatenate([pt - d * L, d], 1))
    # through vertices / along edges, exactly representable targets
    k = rng.integers(0, n, n_each)
    o = rng.uniform(-3, 3, (n_each, 3)).astype(np.float32).astype(np.float64)
    tgt = P[k, rng.integers(0, 3, n_each)].astype(np.float64)
    d = tgt - o
    rays.append(np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1))
    # axis-parallel and nearly axis-parallel (huge |1/d| on two axes)
    o = rng.uniform(-2, 2, (n_each, 3)); d = np.zeros((n_each, 3)); ax = rng.integers(0, 3, n_each)
    d[np.arange(n_each), ax] = rng.choice([-1.0, 1.0], n_each)
    d += rng.choice([0.0, 1e-30, 1e-12, 1e-7, 1e-4], (n_each, 1)) * rng.normal(size=(n_each, 3))
    rays.append(np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1))
    # origins far away (|S| dominates the margin), unnormalised directions
    o = rng.uniform(-1, 1, (n_each, 3)) * 300.0
    d = (rng.uniform(-2, 2, (n_each, 3)) - o) * rng.uniform(0.01, 5.0, (n_each, 1))
    rays.append(np.concatenate([o, d], 1))
    return np.concatenate(rays).astype(np.float32)


@pytest.fixture(scope="module")
def nasty():
    rng = np.random.default_rng(2024)
    P = _nasty_triangles(rng)
    tri, nodes = _build(_tri_array(P))
    return P, tri, nodes, _nasty_rays(P, rng)


def test_pruning_modes_equal_the_unpruned_oracle_on_adversarial_geometry(hip, oracle, nasty):
    P, tri, nodes, rays = nasty
    so = oracle.scene_create(tri, nodes)
    to, do = so.query_hits(rays)
    assert 0.2 < (to >= 0).mean() < 0.99
    info = None
    for mode in (0, 1, 2):
        sg = hip.scene_create(tri, nodes)
        sg.set_option("audit_via_queue", 1)
        sg.set_option("prune", mode)
        info = sg.prune_info()
        assert info["mode"] == mode                      # the scene IS prunable; slivers only flag the records above them
        tg, dg = sg.query_hits(rays)
        assert np.array_equal(tg, to), "mode %d: %d triangle ids differ" % (mode, int((tg != to).sum()))
        assert np.array_equal(_bits(dg), _bits(do))
        sg.set_option("audit_via_queue", 2)              # the primary stage's shared-origin / pre-translated-box variant
        cam = rays[:50000].copy()
        cam[:, :3] = np.float32([0.3, -0.2, 3.5])
        tg, dg = sg.query_hits(cam)
        t2, d2 = so.query_hits(cam)
        assert np.array_equal(tg, t2) and np.array_equal(_bits(dg), _bits(d2))
    assert info["unprunable_triangles"] >= 100           # the slivers were recognised


def test_pruned_frames_equal_the_oracle_on_adversarial_geometry(hip, oracle, nasty):
    P, tri, nodes, _ = nasty
    hdr = scenes.synthetic_hdr(64, 32)
    so = oracle.scene_create(tri, nodes)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(20, 10, 6)
    p = trace.make_params(200, 150, eye, cam, 50, 4, spp=3)
    want = so.render(p)
    for mode in (0, 1, 2):
        sg = hip.scene_create(tri, nodes)
        sg.set_env(hdr, None, 1)
        sg.set_option("prune", mode)
        assert np.array_equal(_bits(sg.render(p)), _bits(want)), mode


def test_stack_overflow_of_the_nearest_first_order_goes_to_the_redo_list(hip, oracle, bunny_small):
    """Nearest-first traversal has no small worst-case stack bound; a ray that needs more rows than the launch has is
    re-traced by the in-order kernel.  debug_stack_cap forces that route for most rays."""
    so = bunny_small.upload(oracle)
    eye, cam = S.camera(10, 5, 3)
    for integ, mb in ((50, 3), (51, 2)):
        p = trace.make_params(128, 96, eye, cam, integ, mb, spp=2)
        want = so.render(p)
        for cap in (1, 2, 4):
            sg = bunny_small.upload(hip)
            sg.set_option("prune", 2)
            sg.set_option("debug_stack_cap", cap)
            assert np.array_equal(_bits(sg.render(p)), _bits(want)), (integ, cap)
    sg = bunny_small.upload(hip)
    sg.set_option("prune", 2)
    sg.set_option("debug_stack_cap", 2)
    sg.set_option("audit_via_queue", 1)
    tg, dg, _ = sg.render_paths(trace.make_params(128, 96, eye, cam, 51, 2, frame0=1))
    to, do, _ = so.render_paths(trace.make_params(128, 96, eye, cam, 51, 2, frame0=1))
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


def test_scenes_outside_the_bound_are_traced_unpruned(hip, oracle, bunny_small):
    """A leaf box that does not hold its triangles (caller arrays): nothing may be skipped on its account."""
    nodes = bunny_small.nodes.copy()
    leaves = np.nonzero(nodes[:, 3] > 0)[0]
    rng = np.random.default_rng(3)
    pick = rng.choice(leaves, 40, replace=False)
    c = (nodes[pick, 6:9] + nodes[pick, 9:12]) * np.float32(0.5)
    nodes[pick, 6:9] = c + (nodes[pick, 6:9] - c) * np.float32(0.5)     # shrunk leaf boxes (still nested in their parents)
    nodes[pick, 9:12] = c + (nodes[pick, 9:12] - c) * np.float32(0.5)
    sg, so = hip.scene_create(bunny_small.tri, nodes), oracle.scene_create(bunny_small.tri, nodes)
    assert sg.prune_info()["mode"] == -1
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(10, 5, 3)
    p = trace.make_params(160, 120, eye, cam, 50, 3, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))


def test_prune_info_of_the_bunny_scene(hip, bunny_small):
    info = bunny_small.upload(hip).prune_info()
    assert info["mode"] == 2                       # the default
    assert 0 < info["margin_a"] < 1e-3             # small against a scene of ~20 units
    assert info["unprunable_triangles"] >= 4       # the thin side faces of the flattened floor box
