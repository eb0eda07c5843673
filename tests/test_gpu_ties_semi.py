"""Two routes that used to end on the redo list -- one lane of the in-order kernel per ray -- and now stay in the 4-wide
kernel (round 3), checked through the TIMED kernels against the oracle:

* exact ties in t are ordered at the two leaves' lowest common ancestor in the REFERENCE's tree (tie_precedes: near child
  first, ties right-first, strict < keeps the first found -- P5/fsh:247, 274, 291-298); a third candidate at the same distance
  still goes to the redo list;
* rays with a direction component that is exactly +-0 (1/d = +-inf) are traversed with a watch for 0 x inf = NaN in the slab
  products (knob semi; on for the MIS integrators' bounce stages, whose env shadow rays have L.z = 0 for every cache cell
  with x = 0.5).
"""
import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _copies(bunny_small, k, seed):
    """every 7th triangle of the Bunny scene, k identical copies each (different colours), shuffled: every hit is a k-way tie"""
    base = bunny_small.tri[:5300:7]
    parts = []
    for c in range(k):
        t = base.copy()
        t[:, 21:24] = (0.9 - 0.3 * c, 0.1 + 0.3 * c, 0.1)
        parts.append(t)
    tri = np.concatenate(parts)
    tri = tri[np.random.default_rng(seed).permutation(tri.shape[0])]
    hs = S.HostScene()
    hs.addTriangles(tri)
    hs.buildBVHwithSAH(8)
    return hs.encode()


@pytest.mark.parametrize("k", [2, 3, 4])
def test_k_way_ties_keep_the_reference_winner(hip, oracle, k):
    bs = scenes.bunny_scene(subdiv=0)
    tri, nodes = _copies(bs, k, 20 + k)
    so = oracle.scene_create(tri, nodes)
    hdr = scenes.synthetic_hdr(64, 32)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(15, 10, 4)
    p = trace.make_params(160, 120, eye, cam, 50, 3, spp=2)
    want = so.render(p)
    pa = trace.make_params(160, 120, eye, cam, 50, 3, frame0=1)
    to, do, _ = so.render_paths(pa)
    assert (to[..., 0] >= 0).mean() > 0.02
    for opts in ({}, {"tie_lca": 0}, {"prune": 0}, {"steal": 0}, {"leaf_threshold": 1}, {"leaf_threshold": 64}):
        sg = hip.scene_create(tri, nodes)
        sg.set_env(hdr, None, 1)
        for kk, v in opts.items():
            sg.set_option(kk, v)
        assert np.array_equal(_bits(sg.render(p)), _bits(want)), (k, opts)
        sg.set_option("audit_via_queue", 1)
        tg, dg, _ = sg.render_paths(pa)
        assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do)), (k, opts)


def test_rays_with_zero_direction_components_in_the_wide_kernel(hip, oracle, bunny_small):
    """semi = 2: every launch keeps such rays.  Origins ON box planes of the zero axis (vertex coordinates are box planes) make
    0 x inf = NaN turn up: those rays must leave for the in-order kernel and still get the reference's answer."""
    rng = np.random.default_rng(31)
    n = 200000
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.uniform(-2, 2, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], 1).astype(np.float32)
    ax = rng.integers(0, 3, n)
    zero = rng.random(n) < 0.5
    rays[zero, 3 + ax[zero]] = rng.choice(np.float32([0.0, -0.0]), int(zero.sum()))
    two = rng.random(n) < 0.1                                     # two zero components: axis-parallel
    rays[two, 3 + (ax[two] + 1) % 3] = 0.0
    # origins whose coordinate on a zero axis IS a vertex coordinate (a box plane)
    P = bunny_small.tri[:, :9].reshape(-1, 3, 3)
    onp = zero & (rng.random(n) < 0.3)
    v = P[rng.integers(0, P.shape[0], n), rng.integers(0, 3, n)]
    rays[onp, ax[onp]] = v[onp, ax[onp]]
    so = bunny_small.upload(oracle)
    to, do = so.query_hits(rays)
    assert (to >= 0).mean() > 0.05
    for semi in (0, 2):
        for prune in (0, 2):
            sg = bunny_small.upload(hip)
            sg.set_option("semi", semi)
            sg.set_option("prune", prune)
            sg.set_option("audit_via_queue", 1)
            tg, dg = sg.query_hits(rays)
            assert np.array_equal(tg, to), (semi, prune, int((tg != to).sum()))
            assert np.array_equal(_bits(dg), _bits(do))


def test_mis_frames_with_the_semi_route_on_and_off(hip, oracle, bunny_small):
    """Chapter 5's env shadow rays: SampleHdr's phi is exactly 0 wherever the cache's x is 0.5, so L.z = 0 in numbers."""
    so = bunny_small.upload(oracle)
    eye, cam = S.camera(10, 5, 3)
    p = trace.make_params(192, 144, eye, cam, 51, 3, spp=3)
    want = so.render(p)
    for semi in (0, 1, 2):
        sg = bunny_small.upload(hip)
        sg.set_option("semi", semi)
        assert np.array_equal(_bits(sg.render(p)), _bits(want)), semi
