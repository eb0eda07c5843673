"""BASELINE.json configs 3 and 4 at their full resolution on the GPU (C3 = Disney
material grid, 512 012 triangles, integrator 4, NEAREST env; C4 = P5 importance sampling + MIS on the
~70k Bunny scene, BILINEAR env + cache).  The oracle cannot render these at full spp in seconds, so
parity is checked on crops / few frames and the full size through the size-independent properties:
multi-chunk rendering (> 2^24 pixel-samples per call), frame-range splits and tile shards."""
import numpy as np
import pytest

from ezrt_amd import scene as S
from ezrt_amd import scenes, trace

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def c3():
    return scenes.disney_grid_scene(subdiv=3)


@pytest.fixture(scope="module")
def c4():
    return scenes.p5_scene(subdiv=2)


@pytest.fixture(scope="module")
def c5():
    return scenes.mega_scene()


def _same_up_to_nan_payload(a, b):
    """Bit equality, except that a NaN equals a NaN (include/ezrt.h: sign and payload of a NaN are not part of the contract)."""
    return bool(((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all())


def test_c3_full_frame_equals_the_oracle(hip, oracle, c3):
    """C3 at its FULL resolution (1024x1024, 512 012 triangles, integrator 4, 4 bounces, NEAREST env), 2 spp: every
    pixel of the frame against the oracle, on the bits (this used to be a hand-run script)."""
    cfg = scenes.CONFIGS["C3"]
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=2)
    got, want = c3.upload(hip).render(p), c3.upload(oracle).render(p)
    assert _same_up_to_nan_payload(got, want)
    assert np.isfinite(want[..., :3]).mean() > 0.999 and float(want[..., :3][np.isfinite(want[..., :3])].max()) > 0.5


def test_c5_full_frame_equals_the_oracle(hip, oracle, c5):
    """C5 at its FULL resolution (2048x2048, 10^6 triangles, integrator 51, 8 bounces = Sobol dims wrap), 1 spp: every
    pixel against the oracle; NaN == NaN (chapter 5's MIS weights can be 0/0, as in the reference)."""
    cfg = scenes.CONFIGS["C5"]
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=1)
    got, want = c5.upload(hip).render(p), c5.upload(oracle).render(p)
    assert _same_up_to_nan_payload(got, want)
    assert np.isfinite(want[..., :3]).mean() > 0.999 and float(want[..., :3][np.isfinite(want[..., :3])].max()) > 0.5


@pytest.fixture(scope="module")
def c2():
    return scenes.bunny_scene(subdiv=2, hdr="shipped")


def _full_frame_at_baseline_spp(hip, oracle, built, name, spp):
    """A COMPLETE frame of a BASELINE config at the stated spp: every pixel of libezrt_hip.so's frame against the CPU oracle's,
    on the bits (NaN == NaN), and the library's own count of non-finite pixels (ezrt_frame_nonfinite) against numpy's."""
    import torch
    cfg = scenes.CONFIGS[name]
    eye, cam = S.camera(*cfg["camera"])
    W, H = cfg["width"], cfg["height"]
    p = trace.make_params(W, H, eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
    sg = built.upload(hip)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    sg.render_device(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    n_bad = hip.frame_nonfinite(acc.data_ptr(), W, H, torch.cuda.current_stream().cuda_stream)
    got = acc.cpu().numpy()
    want = built.upload(oracle).render(p)
    assert _same_up_to_nan_payload(got, want), "%s at %d spp: %d pixels differ" % (name, spp, int((_bits(got) != _bits(want)).any(axis=2).sum()))
    assert n_bad == int((~np.isfinite(want[..., :3]).all(axis=2)).sum())
    assert np.isfinite(want[..., :3]).mean() > 0.999 and float(want[..., :3][np.isfinite(want[..., :3])].max()) > 0.5
    return n_bad


def test_c2_complete_frame_at_baseline_spp_equals_the_oracle(hip, oracle, c2):
    """BASELINE configs[1], the headline workload: 512x512, integrator 50, 4 bounces, 64 spp -- the whole frame (this was a
    hand-run script, tests/check_config_parity.py; oracle ~1 s on 16 cores)."""
    assert _full_frame_at_baseline_spp(hip, oracle, c2, "C2", 64) == 0


def test_c4_complete_frame_at_baseline_spp_equals_the_oracle(hip, oracle, c4):
    """BASELINE configs[3]: 1024x1024, integrator 51 (env importance sampling + MIS), 2 bounces, 256 spp -- the whole frame
    (oracle ~22 s on 16 cores).  Chapter 5's MIS weights can be 0/0, as in the reference: such pixels are NaN on both sides."""
    _full_frame_at_baseline_spp(hip, oracle, c4, "C4", 256)


def test_c3_complete_frame_at_baseline_spp_equals_the_oracle(hip, oracle, c3):
    """BASELINE configs[2]: 1024x1024, 512 012 triangles, integrator 4, 4 bounces, 128 spp -- the whole frame (oracle ~27 s on 16 cores)."""
    assert _full_frame_at_baseline_spp(hip, oracle, c3, "C3", 128) == 0


def test_c5_complete_frame_at_16_spp_equals_the_oracle(hip, oracle, c5):
    """BASELINE configs[4]: 2048x2048, 10^6 triangles, integrator 51, 8 bounces -- the whole frame at 16 of its 512 spp (the oracle
    needs ~38 s per 16 spp on 16 cores; the 512-spp figure is checked on a crop by bench.py's `configs` block)."""
    _full_frame_at_baseline_spp(hip, oracle, c5, "C5", 16)


def test_ring_stack_spills_do_not_change_a_deep_scene(hip, c5):
    """Round 6: traceq4_kernel's LDS traversal stack is a ring of 16 rows per lane whose oldest entries spill to global memory
    (TraceQ4Args::stack_cap).  On the 10^6-triangle scene -- the deepest tree of the BASELINE configs: rays use up to 17 rows -- a ring
    of 4 rows (every ray spills, most of them repeatedly), a ring of 8, and a ring of 4 with a spill area of 4 entries (spills AND
    hand-overs to the redo list) must give the default's frame on the bits: 8 bounces, env shadow rays, zero-component rays and all."""
    cfg = scenes.CONFIGS["C5"]
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=2, rect=(896, 1024, 1152, 1216))
    want = c5.upload(hip).render(p)
    assert np.isfinite(want[..., :3]).mean() > 0.99 and float(np.nanmax(want[..., :3])) > 0.5
    # (prune_mis 1 / 0: the two-ray launches run a slot-order instance on ABSOLUTE rows while the scene's other launches use the ring --
    # the launch must then allocate the slot-order bound, 21 rows here, not the ring's 17)
    for opts in ({"stack_cap": 4}, {"stack_cap": 8}, {"debug_stack_cap": 2}, {"stack_cap": 32}, {"prune_mis": 1}, {"prune_mis": 0}, {"prune_mis": 1, "stack_cap": 4}):
        sg = c5.upload(hip)
        for k, v in opts.items():
            sg.set_option(k, v)
        assert _same_up_to_nan_payload(sg.render(p), want), opts


def test_c3_disney_grid_full_resolution(hip, oracle, c3):
    assert c3.tri.shape[0] == 25 * 20480 + 12
    sg = c3.upload(hip)
    eye, cam = S.camera(0, 15, 8)
    W = H = 1024
    # 20 spp of 1024^2 = 21 M pixel-samples: two internal chunks
    full = sg.render(trace.make_params(W, H, eye, cam, 4, 4, spp=20))
    assert np.isfinite(full).all() and full[..., :3].max() > 0.5
    part = sg.render(trace.make_params(W, H, eye, cam, 4, 4, spp=7))
    part = sg.render(trace.make_params(W, H, eye, cam, 4, 4, spp=13, frame0=7), part)
    assert np.array_equal(_bits(full), _bits(part))
    img = np.zeros((H, W, 4), np.float32)
    for r in range(4):
        sg.render(trace.make_params(W, H, eye, cam, 4, 4, spp=20, tile=(16, 16), shard=(r, 4)), img)
    assert np.array_equal(_bits(full), _bits(img))
    # oracle on a crop over the spheres, 2 frames, with the traversal counters
    so = c3.upload(oracle)
    rect = (480, 380, 544, 428)
    pc = trace.make_params(W, H, eye, cam, 4, 4, spp=2, rect=rect)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.counters_reset()
    a = sg.render(pc, np.zeros((H, W, 4), np.float32))
    b = so.render(pc, np.zeros((H, W, 4), np.float32))
    assert np.array_equal(_bits(a), _bits(b))
    assert sg.counters() == so.counters()
    hit = a[rect[1]:rect[3], rect[0]:rect[2], :3]
    assert hit.max() > 0.2


def test_c4_mis_full_resolution(hip, oracle, c4):
    sg = c4.upload(hip)
    eye, cam = S.camera(90, 10, 2)  # P5 preset (P5/main.cpp:796-798)
    W = H = 1024
    full = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=18))
    assert np.isfinite(full).all()
    part = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=16))
    part = sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=2, frame0=16), part)
    assert np.array_equal(_bits(full), _bits(part))
    img = np.zeros((H, W, 4), np.float32)
    for r in range(8):
        sg.render(trace.make_params(W, H, eye, cam, 51, 2, spp=18, tile=(16, 16), shard=(r, 8)), img)
    assert np.array_equal(_bits(full), _bits(img))
    so = c4.upload(oracle)
    rect = (470, 420, 560, 500)
    pc = trace.make_params(W, H, eye, cam, 51, 2, spp=3, rect=rect)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.counters_reset()
    a = sg.render(pc, np.zeros((H, W, 4), np.float32))
    b = so.render(pc, np.zeros((H, W, 4), np.float32))
    assert np.array_equal(_bits(a), _bits(b))
    assert sg.counters() == so.counters()
    # audit every ray of the crop (primary, env shadow rays, bounce rays)
    pa = trace.make_params(W, H, eye, cam, 51, 2, frame0=1, rect=rect)
    tg, dg, cg = sg.render_paths(pa)
    to, do, co = so.render_paths(pa)
    sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
    assert np.array_equal(tg[sl], to[sl]) and np.array_equal(_bits(dg[sl]), _bits(do[sl]))
    assert np.array_equal(_bits(cg[sl]), _bits(co[sl]))
    assert (tg[sl][..., 0] >= 0).mean() > 0.3


def test_megakernel_and_streaming_forms_agree(hip, c4):
    """ezrt_render (streaming pipeline) and ezrt_render_paths (megakernel) are two schedules of the
    same arithmetic: one frame rendered by both must give the same sample radiance."""
    sg = c4.upload(hip)
    eye, cam = S.camera(90, 10, 2)
    for integ, mb in ((51, 2), (50, 4)):
        p = trace.make_params(320, 200, eye, cam, integ, mb, spp=1, frame0=0)
        img = sg.render(p)
        _, _, col = sg.render_paths(p)
        assert np.array_equal(_bits(img[..., :3]), _bits(col))


def test_c5_million_triangles_eight_bounces(hip, oracle, c5):
    """C5: exactly 10^6 triangles, 2048^2, integrator 51 with 8 bounces (Sobol dims wrap d & 7).
    Full resolution at reduced spp through the size-independent properties, oracle on a crop."""
    assert c5.tri.shape[0] == 1_000_000
    # (scenes of this size are built by the GPU SAH builder by default since round 4 -- the same arrays as the host's,
    # tests/test_gpu_lbvh.py; the host builder's statistics, incl. that the INF = 114514 cap is live here, are asserted there)
    assert "gpu_build_ms" in c5.build_stats or c5.build_stats["inf_cap_nodes"] > 0
    cfg = scenes.CONFIGS["C5"]
    sg = c5.upload(hip)
    eye, cam = S.camera(*cfg["camera"])
    W, H, mb = cfg["width"], cfg["height"], cfg["max_bounce"]
    full = sg.render(trace.make_params(W, H, eye, cam, 51, mb, spp=6))   # 25 M pixel-samples: 2 chunks
    assert np.isfinite(full).all() and full[..., :3].max() > 0.5
    part = sg.render(trace.make_params(W, H, eye, cam, 51, mb, spp=4))
    part = sg.render(trace.make_params(W, H, eye, cam, 51, mb, spp=2, frame0=4), part)
    assert np.array_equal(_bits(full), _bits(part))
    img = np.zeros((H, W, 4), np.float32)
    for r in range(8):
        sg.render(trace.make_params(W, H, eye, cam, 51, mb, spp=6, tile=(16, 16), shard=(r, 8)), img)
    assert np.array_equal(_bits(full), _bits(img))
    so = c5.upload(oracle)
    rect = (1000, 700, 1096, 764)
    pc = trace.make_params(W, H, eye, cam, 51, mb, spp=2, rect=rect)
    sg.set_instrumentation(1)
    so.set_instrumentation(1)
    sg.counters_reset()
    a = sg.render(pc, np.zeros((H, W, 4), np.float32))
    b = so.render(pc, np.zeros((H, W, 4), np.float32))
    assert np.array_equal(_bits(a), _bits(b))
    assert sg.counters() == so.counters()
    sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
    tg, dg, _ = sg.render_paths(trace.make_params(W, H, eye, cam, 51, mb, frame0=1, rect=rect))
    to, do, _ = so.render_paths(trace.make_params(W, H, eye, cam, 51, mb, frame0=1, rect=rect))
    assert np.array_equal(tg[sl], to[sl]) and np.array_equal(_bits(dg[sl]), _bits(do[sl]))
    assert (tg[sl][..., 0] >= 0).mean() > 0.5
