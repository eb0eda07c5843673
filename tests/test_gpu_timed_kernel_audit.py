"""Direct audit of the TIMED kernels' integers (VERDICT r1 task 1).

`audit_via_queue` routes ezrt_query_hits and ezrt_render_paths through the kernels a render call
times -- traceq_kernel with the render call's template (<false, 6>: cooperative leaves), LDS layout,
static + dynamic pools, intra-wave work stealing with the 64-bit atomicMin merge, the tie redo launch,
and the streaming shading stages that produce each bounce's ray queue -- and returns what they left in
the hit records: {triangle id, t} per ray slot.  Everything is compared with the oracle bit for bit
(P5/fsh:254-306 hitBVH + 238-251 hitArray).
"""
import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _rays(n, seed, lo=-3, hi=3):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3))
    tgt = rng.uniform(-1, 1, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


def _camera_rays(w, h, eye, cam, seed):
    """primary rays of a w x h frame with a random sub-pixel jitter, all from `eye`"""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    px = ((xs + rng.random((h, w))) / w * 2 - 1).astype(np.float32)
    py = ((ys + rng.random((h, w))) / h * 2 - 1).astype(np.float32)
    m = np.asarray(cam, np.float32).reshape(4, 4).T          # column-major -> rows
    v = np.stack([px, py, np.full_like(px, -1.5)], -1)
    d = v @ m[:3, :3].T
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.zeros((h * w, 6), np.float32)
    rays[:, :3] = eye
    rays[:, 3:] = d.reshape(-1, 3)
    return rays


def _check_paths(sg, so, p, min_hit=0.01):
    tg, dg, cg = sg.render_paths(p)
    to, do, co = so.render_paths(p)
    assert np.array_equal(tg, to), "triangle ids differ in %d slots" % int((tg != to).sum())
    assert np.array_equal(_bits(dg), _bits(do))
    assert np.array_equal(_bits(cg), _bits(co))
    assert (to[..., 0] >= 0).mean() > min_hit
    return to


@pytest.mark.parametrize("instr", [0, 1])
def test_query_rays_through_the_timed_kernel(hip, oracle, bunny_small, instr):
    """caller rays as ONE trace stage: random rays incl. axis-parallel ("wild") ones, enough of them
    (2^20) that the static pools, the dynamic pools and the stealing tail all run."""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    sg.set_option("audit_via_queue", 1)
    sg.set_instrumentation(instr)            # 0: <false, 6> cooperative leaves; 1: the counting template
    so.set_instrumentation(instr)
    rays = _rays(1 << 20, 21)
    rng = np.random.default_rng(22)
    wild = rng.random(rays.shape[0]) < 0.02
    rays[wild, 3 + rng.integers(0, 3, wild.sum())] = 0.0
    sg.counters_reset()
    so.counters_reset()
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert (to >= 0).mean() > 0.2
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
    assert sg.counters() == so.counters()
    for n in (1, 63, 64, 65, 5000):           # short queues: fewer rays than waves
        tg, dg = sg.query_hits(rays[:n])
        assert np.array_equal(tg, to[:n]) and np.array_equal(_bits(dg), _bits(do[:n]))


def test_shared_origin_rays_through_the_primary_stage_variant(hip, oracle, bunny_small):
    """audit_via_queue = 2: const_origin + boxes pre-translated by the eye (inner_rel), the variant the
    primary stage of every render call runs"""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    sg.set_option("audit_via_queue", 2)
    for cam_args in ((0, 0, 4), (90, 10, 2), (33, -20, 1.2)):
        eye, cam = S.camera(*cam_args)
        rays = _camera_rays(640, 480, eye, cam, 5)
        tg, dg = sg.query_hits(rays)
        to, do = so.query_hits(rays)
        assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
        assert (to >= 0).mean() > 0.3
    sg.set_option("rel_boxes", 0)
    tg, dg = sg.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


@pytest.mark.parametrize("integ,mb", [(3, 2), (4, 4), (50, 4), (51, 2), (51, 3)])
@pytest.mark.parametrize("instr", [0, 1])
def test_every_stage_queue_of_a_frame_matches_the_oracle(hip, oracle, bunny_small, integ, mb, instr):
    """render_paths served from the streaming pipeline's hit records: every ray slot of every path of a
    frame (primary, env shadow rays, bounce rays), ids and distances bit-equal, -2 where no ray was shot"""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    sg.set_option("audit_via_queue", 1)
    sg.set_instrumentation(instr)
    eye, cam = S.camera(15, 8, 3.0)
    for frame0 in (0, 7):
        p = trace.make_params(200, 152, eye, cam, integ, mb, frame0=frame0, env_clamp=10.0 if integ == 3 else 0.0)
        to = _check_paths(sg, so, p, 0.2)
        assert (to[..., -1] >= -1).any()       # some path reaches the last bounce
    # a pixel rect and a tile shard of the same frame: non-owned pixels keep the caller's values
    p = trace.make_params(200, 152, eye, cam, integ, mb, frame0=3, rect=(40, 30, 170, 140), tile=(16, 16), shard=(1, 3))
    _check_paths(sg, so, p, 0.0)


def test_tie_scene_through_the_timed_kernel(hip, oracle, bunny_small):
    """every triangle duplicated: each hit is an exact tie in t between two ids; the winner follows
    the reference's visit order, so stolen subtrees must be detected and re-traced (redo launch)"""
    twin = bunny_small.tri[:5300:7].copy()
    twin[:, 21:24] = (0.9, 0.1, 0.1)
    tri = np.concatenate([bunny_small.tri[:5300:7], twin])
    rng = np.random.default_rng(12)
    tri = tri[rng.permutation(tri.shape[0])]
    hs = S.HostScene()
    hs.addTriangles(tri)
    hs.buildBVHwithSAH(8)
    t2, n2 = hs.encode()
    sg, so = hip.scene_create(t2, n2), oracle.scene_create(t2, n2)
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    sg.set_option("audit_via_queue", 1)
    eye, cam = S.camera(0, 0, 4)
    _check_paths(sg, so, trace.make_params(128, 128, eye, cam, 50, 3, frame0=1), 0.02)
    rays = _rays(200000, 31)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
    # few rays, so that most lanes steal: the tail regime of a late bounce
    tg, dg = sg.query_hits(rays[:3000])
    assert np.array_equal(tg, to[:3000]) and np.array_equal(_bits(dg), _bits(do[:3000]))


def test_deep_skewed_tree_through_the_timed_kernel(hip, oracle):
    rng = np.random.default_rng(4)
    n = 3000
    T = np.zeros((n, 36), np.float32)
    c = np.stack([np.linspace(-3, 3, n), rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n)], 1)
    P = (c[:, None, :] + rng.uniform(-0.05, 0.05, (n, 3, 3))).astype(np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 0, 1], 3)
    T[:, 18:36] = S.Material.disney(baseColor=(0.8, 0.6, 0.4)).to18()
    hs = S.HostScene()
    hs.addTriangles(T)
    hs.buildBVH(1)
    tri, nodes = hs.encode()
    sg, so = hip.scene_create(tri, nodes), oracle.scene_create(tri, nodes)
    assert sg.stats()["depth"] >= 12
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    sg.set_option("audit_via_queue", 1)
    eye, cam = S.camera(30, 5, 6)
    _check_paths(sg, so, trace.make_params(160, 96, eye, cam, 50, 3, frame0=0), 0.01)
    rays = _rays(100000, 9) * np.array([2, 0.1, 0.1, 1, 1, 1], np.float32)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


def test_full_c2_frame_all_rays_of_a_step(hip, oracle):
    """BASELINE configs[1] at full size (79 820 triangles, 512x512, integrator 50, 4 bounces): all rays
    of complete frames -- what one 1/64th of a bench step traces -- through the timed kernels, ids and t
    equal to the oracle's for every ray slot of every pixel."""
    bs = scenes.bunny_scene(subdiv=2, hdr="shipped")
    sg, so = bs.upload(hip), bs.upload(oracle)
    sg.set_option("audit_via_queue", 1)
    eye, cam = S.camera(0, 0, 4)
    n_rays = 0
    for frame0 in (0, 63):
        p = trace.make_params(512, 512, eye, cam, 50, 4, frame0=frame0)
        to = _check_paths(sg, so, p, 0.3)
        n_rays += int((to >= -1).sum())
    assert n_rays > 2 * 350000      # ~397 k rays per frame on this camera
    # the Bunny-filling P5 preset camera (P5/main.cpp:796-798)
    eye, cam = S.camera(90, 10, 2)
    _check_paths(sg, so, trace.make_params(512, 512, eye, cam, 50, 4, frame0=5), 0.6)
