"""Parity tests proper: the gfx950 HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json north_star): hit-triangle ids bit-exact, radiance within
1e-4 per-channel L-inf -- the arithmetic contract actually makes both bit-exact, which
the tests report and assert where stated."""
import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu

TOL = 1e-4  # per-channel L-inf on radiance, BASELINE.json north_star
INF = np.float32(114514.0)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_hip_library_is_the_backend(hip):
    assert hip.backend() == "hip:gfx950"


def test_detmath_is_bit_identical_on_host_and_device(hip, oracle):
    """The arithmetic contract: + - * / sqrt, int<->float and the det-math built-ins give the same
    bits on x86-64 and gfx950."""
    rng = np.random.default_rng(0)
    n = 1 << 18
    x = rng.uniform(-7, 7, n).astype(np.float32)
    y = rng.uniform(-7, 7, n).astype(np.float32)
    u = rng.uniform(-1.0001, 1.0001, n).astype(np.float32)
    pos = np.exp(rng.uniform(-30, 30, n)).astype(np.float32)
    small = np.concatenate([rng.uniform(0, 1e-37, n // 2), rng.uniform(0, 1, n // 2)]).astype(np.float32)
    cases = [(0, x, None), (1, x, None), (2, x, y), (3, u, None), (4, pos, None), (5, x * 10, None),
             (6, np.abs(u) * 0.01 + 1e-6, np.abs(y) / 7), (7, pos, None), (7, small, None), (8, x, y), (8, pos, small + 1e-30),
             (9, rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32), None)]
    for op, a, b in cases:
        g, o = hip.debug_math(op, a, b), oracle.debug_math(op, a, b)
        same = _bits(g) == _bits(o)
        both_nan = np.isnan(g) & np.isnan(o)
        assert (same | both_nan).all(), "op %d: %d mismatches" % (op, int((~(same | both_nan)).sum()))


def test_query_hits_parity(hip, oracle, bunny_small):
    from test_oracle import camera_rays, random_rays
    rays = np.concatenate([camera_rays(20000, 1), random_rays(20000, 2)])
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to)
    assert np.array_equal(_bits(dg), _bits(do))
    assert 0.2 < (tg >= 0).mean() < 0.9


@pytest.mark.parametrize("integ,bounces,clamp", [(3, 2, 10.0), (4, 4, 0.0), (50, 4, 0.0), (51, 2, 0.0)])
def test_path_audit_every_ray_of_every_pixel(hip, oracle, bunny_small, integ, bounces, clamp):
    """Per pixel-sample discrete-decision audit: the triangle id and distance of every ray of the
    path (primary, env shadow rays, bounce rays) and the sample radiance."""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    eye, cam = S.camera(15, 8, 3.2)
    for frame in (0, 5):
        p = trace.make_params(96, 80, eye, cam, integ, bounces, frame0=frame, env_clamp=clamp)
        tg, dg, cg = sg.render_paths(p)
        to, do, co = so.render_paths(p)
        assert np.array_equal(tg, to), "hit triangle ids differ on %d rays" % int((tg != to).sum())
        assert np.array_equal(_bits(dg), _bits(do))
        err = np.abs(cg - co)
        assert np.nanmax(err) < TOL
        assert np.array_equal(_bits(cg), _bits(co)), "radiance not bit-exact: Linf %g" % np.nanmax(err)
        assert (tg[..., 0] >= 0).mean() > 0.3


@pytest.mark.parametrize("integ,bounces,clamp", [(3, 2, 10.0), (4, 4, 0.0), (50, 4, 0.0), (51, 2, 0.0)])
def test_render_parity_and_counters(hip, oracle, bunny_small, integ, bounces, clamp):
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    eye, cam = S.camera(0, 0, 4)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    p = trace.make_params(128, 128, eye, cam, integ, bounces, spp=6, env_clamp=clamp)
    ig, io = sg.render(p), so.render(p)
    assert np.abs(ig - io).max() < TOL
    assert np.array_equal(_bits(ig), _bits(io))
    assert sg.counters() == so.counters()
    # level-0 instrumentation counts the same rays and nothing else
    sg.set_instrumentation(0)
    sg.counters_reset()
    ig0 = sg.render(p)
    assert np.array_equal(_bits(ig0), _bits(ig))
    c0 = sg.counters()
    assert c0["rays"] == so.counters()["rays"] and c0["node_pops"] == 0


def test_p5_scene_mirror_floor_parity(hip, oracle):
    """P5 materials: metallic clear-coated body on a near-mirror floor (GTR1/GTR2 lobes, pow/log)."""
    bs = scenes.p5_scene(subdiv=0)
    sg, so = bs.upload(hip), bs.upload(oracle)
    eye, cam = S.camera(90, 10, 2)
    p = trace.make_params(96, 96, eye, cam, 51, 2, spp=4)
    ig, io = sg.render(p), so.render(p)
    assert np.nanmax(np.abs(ig - io)) < TOL
    assert np.array_equal(_bits(ig), _bits(io))
    tg, dg, _ = sg.render_paths(p)
    to, do, _ = so.render_paths(p)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


def test_cornell_and_median_builder_parity(hip, oracle):
    """C1 scene (12 triangles: root is an inner node over two leaves) and a median-split tree."""
    eye, cam = S.camera(0, 0, 4)
    for bs in (scenes.cornell_scene(), scenes.bunny_scene(subdiv=0, sah=False, leaf_n=3)):
        sg, so = bs.upload(hip), bs.upload(oracle)
        p = trace.make_params(64, 64, eye, cam, 3, 4, spp=2, env_clamp=10.0)
        assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))


def test_single_leaf_scene(hip, oracle):
    """nTri <= leaf size: the root itself is a leaf."""
    bs = scenes.cornell_scene(leaf_n=16)
    assert bs.nodes.shape[0] == 2
    sg, so = bs.upload(hip), bs.upload(oracle)
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(32, 32, eye, cam, 50, 3, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))


def test_decompositions_reproduce_the_full_image(hip, bunny_small):
    """Ragged sizes, pixel rects, frame-range splits and round-robin tile shards are bit-identical
    to one full render (so the N-GPU image equals the 1-GPU image)."""
    sg = bunny_small.upload(hip)
    eye, cam = S.camera(20, 10, 3)
    W, H = 203, 117  # not multiples of 16
    full = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=5))
    part = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=2, frame0=0))
    part = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=3, frame0=2), part)
    assert np.array_equal(_bits(full), _bits(part))
    img = np.zeros((H, W, 4), np.float32)
    sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=5, rect=(0, 0, 77, H)), img)
    sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=5, rect=(77, 0, W, H)), img)
    assert np.array_equal(_bits(full), _bits(img))
    for tile in ((32, 32), (8, 8), (24, 40)):
        img = np.zeros((H, W, 4), np.float32)
        for r in range(8):
            sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=5, tile=tile, shard=(r, 8)), img)
        assert np.array_equal(_bits(full), _bits(img)), tile


def test_empty_work_and_errors(hip, bunny_small):
    sg = bunny_small.upload(hip)
    eye, cam = S.camera()
    img = np.full((16, 16, 4), 7.0, np.float32)
    sg.render(trace.make_params(16, 16, eye, cam, 50, 4, spp=0), img)
    assert (img == 7.0).all()
    sg.render(trace.make_params(16, 16, eye, cam, 50, 4, spp=1, rect=(3, 3, 3, 9)), img)
    assert (img == 7.0).all()
    with pytest.raises(trace.TraceError, match="integrator"):
        sg.render(trace.make_params(8, 8, eye, cam, 7, 2))
    bad = bunny_small.nodes.copy()
    bad[1, 0] = 0
    with pytest.raises(trace.TraceError, match="children"):
        hip.scene_create(bunny_small.tri, bad)
    sc = hip.scene_create(bunny_small.tri, bunny_small.nodes)
    with pytest.raises(trace.TraceError, match="cache"):
        sc.render(trace.make_params(8, 8, eye, cam, 51, 2))


def test_tonemap_parity(hip, oracle):
    rng = np.random.default_rng(3)
    rgba = rng.uniform(0, 8, (4096, 4)).astype(np.float32)
    assert np.array_equal(hip.tonemap(rgba), oracle.tonemap(rgba))


def test_full_size_bunny_70k_properties_and_sampled_parity(hip, oracle):
    """BASELINE.json configs[1] at full size (79 820 triangles, 512x512, 4 bounces): the oracle
    cannot render 64 spp of it in seconds, so (a) tile-shard union == full render, (b) frame split
    == full, (c) traversal counters and a 64x64 crop of 2 frames bit-equal the oracle."""
    bs = scenes.bunny_scene(subdiv=2)
    assert bs.tri.shape[0] == 79488 + 12 + 320
    sg = bs.upload(hip)
    eye, cam = S.camera(0, 0, 4)
    W = H = 512
    full = sg.render(trace.make_params(W, H, eye, cam, 50, 4, spp=8))
    assert np.isfinite(full).all()
    img = np.zeros((H, W, 4), np.float32)
    for r in range(8):
        sg.render(trace.make_params(W, H, eye, cam, 50, 4, spp=8, shard=(r, 8)), img)
    assert np.array_equal(_bits(full), _bits(img))
    part = sg.render(trace.make_params(W, H, eye, cam, 50, 4, spp=3))
    part = sg.render(trace.make_params(W, H, eye, cam, 50, 4, spp=5, frame0=3), part)
    assert np.array_equal(_bits(full), _bits(part))
    so = bs.upload(oracle)
    rect = (224, 150, 288, 214)  # across the bunny
    pc = trace.make_params(W, H, eye, cam, 50, 4, spp=2, rect=rect)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.counters_reset()
    a = sg.render(pc, np.zeros((H, W, 4), np.float32))
    b = so.render(pc, np.zeros((H, W, 4), np.float32))
    assert np.array_equal(_bits(a), _bits(b))
    assert sg.counters() == so.counters()
    crop = full[rect[1]:rect[3], rect[0]:rect[2]]
    assert crop[..., :3].max() > 0.5


def _same_up_to_nan_payload(a, b):
    """Bit equality, except that a NaN equals a NaN (sign and payload of a NaN are not part of the contract)."""
    return bool(((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("integ", [51, 52])
def test_c4_at_full_resolution_equals_the_oracle(hip, oracle, integ):
    """BASELINE.json's C4 (chapter 5's scene, 1024x1024, MIS, 2 bounces) at full resolution and a reduced spp: the MIS
    integrators' stage-1 state, shared ray origins, env planes and multi-megasample chunks, against the oracle on the bits."""
    bs = scenes.p5_scene(subdiv=2, hdr="shipped")
    eye, cam = S.camera(*scenes.CONFIGS["C4"]["camera"])
    p = trace.make_params(1024, 1024, eye, cam, integ, 2, spp=6)
    got, want = bs.upload(hip).render(p), bs.upload(oracle).render(p)
    assert _same_up_to_nan_payload(got, want)
    assert np.isfinite(want[..., :3]).mean() > 0.999 and float(want[..., :3][np.isfinite(want[..., :3])].max()) > 0.5


@pytest.mark.parametrize("integ", [50, 51])
def test_eight_bounces_wrap_the_sobol_table(hip, oracle, bunny_small, integ):
    """BASELINE.json's stress config asks for 8 bounces; the shader's table has 8 dimensions = 4
    bounces.  Defined here: dimensions wrap (d & 7).  GPU == oracle for max_bounce 8."""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    eye, cam = S.camera(10, 5, 2.5)
    p = trace.make_params(72, 56, eye, cam, integ, 8, spp=3)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    tg, dg, _ = sg.render_paths(trace.make_params(72, 56, eye, cam, integ, 8, frame0=2))
    to, do, _ = so.render_paths(trace.make_params(72, 56, eye, cam, integ, 8, frame0=2))
    assert tg.shape[-1] == 17 and np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
    assert (tg[..., 10] >= -1).any()  # some paths really reach the 5th bounce


def test_exact_distance_ties_follow_the_reference_visit_order(hip, oracle, bunny_small):
    """Every triangle duplicated (identical copies => every hit is an exact tie in t between two
    triangle ids that usually sit in different leaves).  The winner is decided by the reference's
    visit order (near child first, ties right-first, strict <): the 4-wide kernel orders a tie at the lowest common
    ancestor of the two leaves (or hands the ray to the in-order kernel); triangle ids must still equal the oracle's."""
    twin = bunny_small.tri[:5300:7].copy()
    twin[:, 21:24] = (0.9, 0.1, 0.1)  # the copy is red: the image shows which twin won each tie
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
    eye, cam = S.camera(0, 0, 4)
    p = trace.make_params(128, 128, eye, cam, 50, 3, spp=3)
    io = so.render(p)
    for opts in ({}, {"tie_lca": 0}, {"wide4": 0}, {"prune": 0}):
        for k, v in opts.items():
            sg.set_option(k, v)
        assert np.array_equal(_bits(sg.render(p)), _bits(io)), opts
        for k in opts:
            sg.set_option(k, {"tie_lca": 1, "wide4": 1, "prune": 2}[k])
    sg.counters_reset()
    sg.render(p)
    so.counters_reset()
    so.render(p)
    assert sg.counters()["rays"] == so.counters()["rays"]
    tg, dg, _ = sg.render_paths(trace.make_params(128, 128, eye, cam, 50, 3, frame0=1))
    to, do, _ = so.render_paths(trace.make_params(128, 128, eye, cam, 50, 3, frame0=1))
    assert np.array_equal(tg, to) and (tg[..., 0] >= 0).mean() > 0.02


def test_fast_reciprocal_is_the_ieee_division_for_every_float(hip):
    """ez_rcp (ezrt_device.h: v_rcp_f32 + one fused Newton step inside [2^-120, 2^120], the compiler's division outside) must be
    `1.0f / x` on the bits: ALL 2^32 patterns are compared on the device (ezrt_debug_math op 18)."""
    out = hip.debug_math(18, np.zeros(2, np.float32))
    assert out[0] == 0.0, "mismatches: %g, first input bits 0x%08x" % (out[0], int(out[1:2].view(np.uint32)[0]))


def test_frame_sizes_that_are_not_powers_of_two(hip, oracle, bunny_small):
    """Frame sizes of every kind through the in-launch ray generation (primary_dir): powers of two and not, W != H."""
    eye, cam = S.camera(20, 10, 3.5)
    for w, h in ((96, 80), (128, 80), (100, 64), (64, 64)):
        p = trace.make_params(w, h, eye, cam, 50, 3, spp=2)
        assert np.array_equal(_bits(bunny_small.upload(hip).render(p)), _bits(bunny_small.upload(oracle).render(p))), (w, h)


def test_schedule_knobs_never_change_results(hip, bunny_small):
    """ezrt_set_option only reschedules the same arithmetic: every combination is bit-identical."""
    sg = bunny_small.upload(hip)
    eye, cam = S.camera(0, 0, 4)
    for integ, mb in ((51, 2), (50, 3)):
        p = trace.make_params(160, 120, eye, cam, integ, mb, spp=3)
        ref = sg.render(p)
        for opts in ({"megakernel": 1}, {"leaf_threshold": 1}, {"leaf_threshold": 64}, {"pool_max": 8},
                     {"pool_div": 16, "pool_max": 2048}, {"trace_wps": 4}, {"trace_wps": 5}, {"trace_wps": 7}, {"lds_nodes": 0},
                     {"lds_nodes": 7}, {"steal": 0}, {"scatter": 0}, {"scatter": 5}, {"scatter": 8}, {"static_pct": 0},
                     {"static_pct": 90}, {"refill_min": 1}, {"refill_min": 64}, {"rel_boxes": 0}, {"wide4": 0},
                     {"wide4": 0, "steal": 0}, {"wide4": 0, "rel_boxes": 0}, {"refill_min": 16}, 
                     {"debug_force_pending": 2}, {"launch_events": 1},
                     {"chunk_log2": 12}, {"chunk_log2": 16}, {"shade_wgs": 1}, {"shade_wgs": 4096}, {"debug_oom_above": 30000},
                     {"debug_oom_above": 45000}, {"env_rgbe": 0}, {"env_planes": 0}, {"trace_wps_rel": 0}, {"trace_wps_rel": 5},
                     {"debug_force_pending": 3}, {"debug_force_pending": 1}, {"debug_force_pending": 5},
                     {"prune": 0}, {"prune": 1}, {"prune": 2}, {"prune": 1, "steal": 0}, {"prune": 2, "debug_stack_cap": 1},
                     {"prune": 2, "debug_stack_cap": 3}, {"prune": 2, "lds_nodes": 0},
                     {"prune": 1, "trace_wps": 4}, {"prune": 2, "prune_min_records": 1 << 20}, {"gen_primary": 0},
                     {"gen_primary": 0, "prune": 0}, {"stack_cap": 4}, {"stack_cap": 9, "prune": 2}, {"min_staged": 0},
                     {"min_staged": 4096}, {"prune_mis": 1}, {"semi": 0}, {"semi": 2}, {"semi": 2, "prune": 0}, {"tie_lca": 0},
                     {"tie_lca": 0, "semi": 0}, {"anyhit": 0}, {"anyhit": 0, "semi": 0}, {"anyhit": 1, "steal": 0},
                     {"anyhit": 1, "debug_force_pending": 3}, {"anyhit": 1, "debug_stack_cap": 1}, {"anyhit": 1, "prune": 0},
                     {"lazy_dir": 0}, {"lazy_dir": 1, "debug_force_pending": 3}, {"refill_min_rel": 0}, {"refill_min_rel": 64},
                     {"refill_min_rel": 1, "refill_min": 1}, {"pipeline_calls": 0}, {"pipeline_calls": 0, "chunk_log2": 12},
                     {"pipeline_calls": 2, "chunk_log2": 12}, {"pipeline_calls": 2, "debug_oom_above": 30000},
                     {"pipeline_calls": 2, "debug_force_pending": 3}, {"bounce_scatter": 0},
                     {"bounce_scatter": 1}, {"bounce_scatter": 1, "debug_force_pending": 3}, {"bounce_scatter": 1, "steal": 0},
                     {"bounce_scatter": 1, "pool_max": 8}, {"bounce_scatter": 1, "static_pct": 0},
                     {"bounce_scatter": 1, "chunk_log2": 12}, {"handover": 0}, {"handover": 1, "steal": 0},
                     {"handover": 1, "debug_force_pending": 3}):
            s2 = bunny_small.upload(hip)
            for k, v in opts.items():
                s2.set_option(k, v)
            assert np.array_equal(_bits(s2.render(p)), _bits(ref)), (integ, opts)
    with pytest.raises(trace.TraceError, match="unknown option"):
        sg.set_option("no_such_knob", 1)
    # scratch that does not fit even for one frame per chunk is an error, not a crash
    s3 = bunny_small.upload(hip)
    s3.set_option("debug_oom_above", 1)
    with pytest.raises(trace.TraceError, match="render scratch"):
        s3.render(p)
    # values that would hang or corrupt a launch are refused: the documented ranges
    for k, v in (("pool_max", 0), ("pool_div", 0), ("lds_nodes", -1), ("leaf_threshold", 0),
                 ("leaf_threshold", 65), ("trace_wps", 0), ("bounce_scatter", 2)):
        with pytest.raises(trace.TraceError, match="outside"):
            sg.set_option(k, v)


def test_small_pools_with_several_static_rounds_draw_every_ray(hip, bunny_small):
    """under the scattered draw of the bounce stages a whole pool can fall into the padding past the queue's end;
    a wave whose FIRST pool did used to retire although its later static rounds held rays (pool_max 8, static_pct 50 on
    a queue of >= 200 000 rays: static_rounds >= 2).  The frame is large enough for that (786 432 pixel-samples in one
    chunk) and every schedule must give the default's bits."""
    eye, cam = S.camera(0, 0, 4)
    for integ, mb in ((50, 3), (51, 2)):
        p = trace.make_params(512, 512, eye, cam, integ, mb, spp=3)
        ref = bunny_small.upload(hip).render(p)
        for opts in ({"bounce_scatter": 1, "pool_max": 8, "pipeline_calls": 0, "static_pct": 50},
                     {"bounce_scatter": 1, "pool_max": 8, "pipeline_calls": 0, "static_pct": 95},
                     {"bounce_scatter": 1, "pool_div": 8, "pipeline_calls": 0},
                     {"bounce_scatter": 1, "pool_div": 8, "pipeline_calls": 0, "static_pct": 90},
                     {"bounce_scatter": 1, "pool_max": 8, "static_pct_pipelined": 50},
                     {"bounce_scatter": 1, "pool_max": 16, "static_pct": 95}, {"handover": 0}):
            s2 = bunny_small.upload(hip)
            for k, v in opts.items():
                s2.set_option(k, v)
            assert np.array_equal(_bits(s2.render(p)), _bits(ref)), (integ, opts)


def test_boxes_that_are_not_nested_fall_back_to_the_binary_kernel(hip, oracle, bunny_small):
    """traceq4_kernel tests descendants without their ancestors, which is only the reference's traversal when
    every box lies inside its parent's (true for the builders' trees).  Caller arrays that violate it -- here
    some inner boxes shrunk so that their children stick out -- must still give the reference's answer."""
    nodes = bunny_small.nodes.copy()
    rng = np.random.default_rng(5)
    inner = np.nonzero(nodes[2:, 3] == 0)[0] + 2
    pick = rng.choice(inner, 60, replace=False)
    c = (nodes[pick, 6:9] + nodes[pick, 9:12]) * np.float32(0.5)
    nodes[pick, 6:9] = c + (nodes[pick, 6:9] - c) * np.float32(0.6)
    nodes[pick, 9:12] = c + (nodes[pick, 9:12] - c) * np.float32(0.6)
    sg, so = hip.scene_create(bunny_small.tri, nodes), oracle.scene_create(bunny_small.tri, nodes)
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(10, 5, 3)
    p = trace.make_params(160, 120, eye, cam, 50, 3, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    sg.set_option("audit_via_queue", 1)
    tg, dg, _ = sg.render_paths(trace.make_params(160, 120, eye, cam, 50, 3, frame0=1))
    to, do, _ = so.render_paths(trace.make_params(160, 120, eye, cam, 50, 3, frame0=1))
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


def test_node_arrays_that_are_a_dag_keep_the_binary_kernel(hip, oracle, bunny_small):
    """Caller arrays may reference an inner node from several parents (validation only asks parent < child).  The
    4-wide collapse indexes its records by node, so such arrays must not reach it (a shared inner node got
    two records, one never numbered -> a write before the record vector's buffer, and chains of shared nodes multiplied
    records).  Here: 200 inner nodes get their RIGHT child replaced by a later inner node that already has a parent --
    still parent < child, boxes enlarged so that they stay nested -- and the frame must be the reference's."""
    nodes = bunny_small.nodes.copy()
    n = nodes.shape[0]
    is_inner = nodes[:, 3] == 0
    is_inner[:2] = False
    rng = np.random.default_rng(11)
    cand = [i for i in np.nonzero(is_inner)[0] if is_inner[int(nodes[i, 1])]]
    parents = rng.choice(cand, 200, replace=False)
    inner_ids = np.nonzero(is_inner)[0]
    for i in parents:
        later = inner_ids[inner_ids > max(int(nodes[i, 0]), int(nodes[i, 1]))]
        if later.size == 0:
            continue
        nodes[i, 1] = np.float32(rng.choice(later))
    # make every box contain its children's again (ids are topologically ordered: one backwards pass)
    for i in range(n - 1, 0, -1):
        if nodes[i, 3] == 0:
            for c in (int(nodes[i, 0]), int(nodes[i, 1])):
                nodes[i, 6:9] = np.minimum(nodes[i, 6:9], nodes[c, 6:9])
                nodes[i, 9:12] = np.maximum(nodes[i, 9:12], nodes[c, 9:12])
    sg, so = hip.scene_create(bunny_small.tri, nodes), oracle.scene_create(bunny_small.tri, nodes)
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(10, 5, 3)
    p = trace.make_params(96, 80, eye, cam, 50, 3, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.render(p), so.render(p)
    assert sg.counters() == so.counters()


def test_deep_skewed_tree_uses_many_stack_rows(hip, oracle):
    """A median-split tree with leaf size 1 over a thin strip of triangles: depth 12+, every ray
    walks long chains; exercises the LDS stack rows and the leaf encoding with n = 1."""
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
    assert sg.stats()["depth"] >= 12 and sg.stats()["max_leaf"] == 1
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(30, 5, 6)
    p = trace.make_params(160, 96, eye, cam, 50, 3, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    from test_oracle import random_rays
    rays = random_rays(20000, 9) * np.array([2, 0.1, 0.1, 1, 1, 1], np.float32)
    tg, dg = sg.query_hits(rays)
    to, do = so.query_hits(rays)
    assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))


def _chain_tree(n_inner):
    """A degenerate tree: inner node k has leaf k (one triangle) on the left and inner node k + 1 on the right; the last
    inner node has two leaves.  Depth = n_inner + 1.  Returns (tri[n,36], nodes[m,12]) in the reference encoding."""
    n = n_inner + 1
    rng = np.random.default_rng(9)
    T = np.zeros((n, 36), np.float32)
    c = np.stack([np.linspace(-3, 3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)], 1)
    P = (c[:, None, :] + rng.uniform(-0.15, 0.15, (n, 3, 3))).astype(np.float32)
    T[:, :9] = P.reshape(n, 9)
    T[:, 9:18] = np.tile([0, 0, 1], 3)
    T[:, 18:36] = S.Material.disney(baseColor=(0.8, 0.6, 0.4)).to18()
    lo = np.minimum.accumulate(P.min(axis=1)[::-1], axis=0)[::-1]   # box of triangles k..n-1
    hi = np.maximum.accumulate(P.max(axis=1)[::-1], axis=0)[::-1]
    nodes = np.zeros((1 + 2 * n_inner + 1, 12), np.float32)
    inner_id = lambda k: 1 + 2 * k          # inner k, followed by its left leaf
    for k in range(n_inner):
        i = inner_id(k)
        last = k == n_inner - 1
        nodes[i] = [i + 1, i + 2, 0, 0, 0, 0, *lo[k], *hi[k]]
        nodes[i + 1] = [0, 0, 0, 1, k, 0, *P[k].min(axis=0), *P[k].max(axis=0)]
        if last:
            nodes[i + 2] = [0, 0, 0, 1, k + 1, 0, *P[k + 1].min(axis=0), *P[k + 1].max(axis=0)]
    return T, nodes


@pytest.mark.gpu
def test_deepest_supported_tree_renders_and_deeper_ones_are_refused_at_create(hip, oracle):
    """The LDS traversal stack holds depth <= 63 (deeper trees used to pass ezrt_scene_create and then fail
    every launch with a generic error)."""
    tri, nodes = _chain_tree(62)              # depth 63
    sg, so = hip.scene_create(tri, nodes), oracle.scene_create(tri, nodes)
    assert sg.stats()["depth"] == 63
    hdr = scenes.synthetic_hdr(64, 32)
    sg.set_env(hdr, None, 1)
    so.set_env(hdr, None, 1)
    eye, cam = S.camera(30, 5, 6)
    p = trace.make_params(96, 64, eye, cam, 50, 2, spp=2)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    for via in (1, 0):
        sg.set_option("audit_via_queue", via)
        tg, dg, _ = sg.render_paths(p)
        to, do, _ = so.render_paths(p)
        assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do))
    sg.set_option("megakernel", 1)
    assert np.array_equal(_bits(sg.render(p)), _bits(so.render(p)))
    tri, nodes = _chain_tree(63)              # depth 64
    with pytest.raises(trace.TraceError, match="depth 64"):
        hip.scene_create(tri, nodes)


@pytest.mark.gpu
def test_invariant_divisor_division_is_exact(hip, oracle):
    """FastDiv (ezrt_device.h): the queue <-> sample slot <-> pixel maps divide by launch-time counts with a multiply
    and two shifts; must equal n / d for EVERY 32-bit n."""
    rng = np.random.default_rng(3)
    edge = np.array([0, 1, 2, 3, 0x7fffffff, 0x80000000, 0x80000001, 0xfffffffe, 0xffffffff], np.uint32)
    for d in [1, 2, 3, 5, 7, 255, 256, 257, 1000, 1024, 4096, 4097, 65535, 65536, 65537, (1 << 20) - 1, 1 << 20, (1 << 20) + 1,
              0x7fffffff, 0x80000000, 0x80000001, 0xffffffff] + [int(x) for x in rng.integers(1, 1 << 22, 40)]:
        n = np.concatenate([edge, rng.integers(0, 1 << 32, 50_000, dtype=np.uint64).astype(np.uint32),
                            (np.arange(0, 3000, dtype=np.uint64) * d).astype(np.uint32),          # multiples of d and their neighbours
                            ((np.arange(1, 3000, dtype=np.uint64) * d - 1) & 0xffffffff).astype(np.uint32)])
        b = np.zeros(n.size, np.uint32)
        b[0] = d
        got = hip.debug_math(17, n.view(np.float32), b.view(np.float32), n=n.size).view(np.uint32)
        assert np.array_equal(got, n // np.uint32(d)), d
        assert np.array_equal(oracle.debug_math(17, n.view(np.float32), b.view(np.float32), n=n.size).view(np.uint32), got)


@pytest.mark.gpu
def test_rgbe_form_of_the_env_map_is_exact(hip, oracle):
    """A map HDRLoader decoded is (m / 256) 2^(E - 128) per channel: the trace then reads it as 4-byte RGBE texels (2 MB
    instead of 8 for 1024x512).  Same images as the float4 path and as the oracle, bilinear and nearest, including a
    map with tiny exponents (subnormal texels), a black texel and unnormalised mantissas; a map that is not RGBE-exact
    silently keeps the float4 path."""
    rng = np.random.default_rng(21)
    rgbe = rng.integers(0, 256, (64, 128, 4), dtype=np.uint8)
    rgbe[..., 3] = rng.integers(118, 140, (64, 128))
    rgbe[0, :8, 3] = np.arange(8)               # E = 0..7: (m / 256) 2^-128 ... subnormal floats
    rgbe[1, :4, :3] = 0                         # black with an arbitrary exponent
    rgbe[2, :16, :3] = rng.integers(0, 16, (16, 3))   # unnormalised: largest mantissa < 128
    env = ((rgbe[..., :3].astype(np.float32) / np.float32(256.0)) *
           np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 128)[..., None]).astype(np.float32)
    shipped = scenes.shipped_hdr()
    eye, cam = S.camera(30, 20, 4)
    for name, hdr in (("random rgbe", env), ("shipped", shipped), ("not rgbe", (env * np.float32(1.0000001) + np.float32(1e-3)).astype(np.float32))):
        bs = scenes.bunny_scene(subdiv=0, hdr=np.ascontiguousarray(hdr), want_cache=True)
        for filt in (1, 0):
            sg, so = bs.upload(hip), bs.upload(oracle)
            sg.set_env(bs.hdr, bs.cache, filt)
            so.set_env(bs.hdr, bs.cache, filt)
            for integ in (50, 51):
                p = trace.make_params(96, 64, eye, cam, integ, 3, spp=2)
                want = so.render(p)
                assert np.array_equal(_bits(sg.render(p)), _bits(want)), (name, filt, integ)
                sg.set_option("env_rgbe", 0)
                assert np.array_equal(_bits(sg.render(p)), _bits(want)), (name, filt, integ, "float4 path")
                sg.set_option("env_rgbe", 1)
                sg.set_option("megakernel", 1)
                assert np.array_equal(_bits(sg.render(p)), _bits(want)), (name, filt, integ, "megakernel")
                sg.set_option("megakernel", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("integ", [3, 4, 50, 51, 52])
def test_pending_rays_take_the_redo_route_under_the_first_shading_pass(hip, oracle, bunny_small, integ):
    """Rays traceq4_kernel hands to the redo list without an answer (not tame) are published as HIT_PENDING and re-traced by
    the stage's redo launch in the reference's order, in line before any shading (until round 6 a knob could run it on a side stream
    under the first shading pass, which is why the shading passes still know how to defer a pending path).  The test hook sends
    every k-th ray slot that way: images, path logs and ray counts must not change."""
    sg, so = bunny_small.upload(hip), bunny_small.upload(oracle)
    eye, cam = S.camera(25, 10, 4)
    p = trace.make_params(128, 96, eye, cam, integ, 3, spp=3, rect=(5, 3, 120, 90))
    want = so.render(p)
    to, do, co = so.render_paths(p)
    rays = so.counters()["rays"]
    for k in (1, 2, 3, 11):
        sg.set_option("debug_force_pending", k)
        sg.counters_reset()
        assert np.array_equal(_bits(sg.render(p)), _bits(want)), (integ, k)
        sg.set_option("audit_via_queue", 1)
        tg, dg, cg = sg.render_paths(p)
        sg.set_option("audit_via_queue", 0)
        assert np.array_equal(tg, to) and np.array_equal(_bits(dg), _bits(do)) and np.array_equal(_bits(cg), _bits(co)), (integ, k)
    assert rays > 0
    sg.set_option("debug_force_pending", 2)
    sg.counters_reset()
    so.counters_reset()
    sg.render(p)
    so.render(p)
    assert sg.counters()["rays"] == so.counters()["rays"] and sg.counters()["samples"] == so.counters()["samples"]


def test_whole_block_frames_read_directions_lazily(hip, oracle, bunny_small):
    """One shard and a frame of whole 16x16 blocks: the primary stage's first shading pass reads a ray's direction only after
    its hit record said "miss" (knob lazy_dir).  Same frames with the knob off, against the oracle, for a plain and a MIS
    integrator; a frame with a ragged edge (not whole blocks) takes the eager path by itself."""
    so = bunny_small.upload(oracle)
    eye, cam = S.camera(10, 5, 3)
    for integ, mb in ((50, 3), (51, 2)):
        for (w, h) in ((160, 128), (160, 120)):
            p = trace.make_params(w, h, eye, cam, integ, mb, spp=2)
            want = so.render(p)
            for lazy in (1, 0):
                sg = bunny_small.upload(hip)
                sg.set_option("lazy_dir", lazy)
                assert np.array_equal(_bits(sg.render(p)), _bits(want)), (integ, w, h, lazy)
