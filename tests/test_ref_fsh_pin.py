"""Pins the SHADER half of the oracle by execution: oracle/ezrt_oracle.c's restatement of the reference's
fragment shaders against oracle/_ref/libezrt_ref_fsh_p{3,4,5}.so = the shaders THEMSELVES (`part {3,4,5} .../shaders/
fshader.fsh`) compiled by g++ through a syntax-only source pass and a GLSL language shim (oracle/ref_recipe/fsh_pass.py,
wrap_fsh.cpp, shim/glsl_shim.h).

Compared on the bits: the per-sample seed, the shaders' own BRDF_Evaluate / SampleBRDF / BRDF_Pdf / hdrPdf / SampleHdr /
hdrColor / hemisphere sampling on 10^5 random inputs, hitBVH's whole HitResult on camera and random rays, and main()'s
running mean per pixel for the integrators 3 / 4 / 50 / 51 over frames 0..5 of a 96x80 view of the shipped-Bunny scene.
What stays "defined here" and is shared by both sides: the precision of sin cos atan asin log pow (ezrt_detmath.h), the
expansion of dot / normalize / mix / min / max / reflect, texture filtering (glsl_shim.h restates the definitions of
DESIGN.md 2).  Which operations run, in which order, on which operands is the reference's text on one side and the
restatement on the other -- draw order (Q9), the mis-named isotropic "aniso" (Q12), sin(elevation) (Q13) included.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref as R  # noqa: E402

from ezrt_amd import _abi, scene as S, scenes, trace  # noqa: E402

pytestmark = pytest.mark.skipif(not all(R.fsh_available(c) for c in (3, 4, 5)),
                                reason="oracle/_ref/libezrt_ref_fsh_p*.so not built (python oracle/ref_recipe/build_ref.py)")
_F = C.POINTER(C.c_float)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit equality; a NaN equals a NaN (include/ezrt.h: sign and payload of a NaN are not part of the contract)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.fixture(scope="module")
def fsh():
    return {c: R.Fsh(c) for c in (3, 4, 5)}


@pytest.fixture(scope="module")
def ofn(oracle):
    oracle.lib.ezrt_oracle_fn.argtypes = [C.c_void_p, C.c_int, C.c_int, _F, _F, C.c_int, _F]
    oracle.lib.ezrt_oracle_fn.restype = C.c_int

    def call(scene, op, chapter, a, b=None):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, R.Fsh.IN_WIDTH[op])
        n = a.shape[0]
        bb = np.ascontiguousarray(b, np.float32) if b is not None else np.zeros((n, 18), np.float32)
        out = np.zeros((n, R.Fsh.OUT_WIDTH[op]), np.float32)
        h = scene._h if scene is not None else None
        rc = oracle.lib.ezrt_oracle_fn(h, op, chapter, a.ctypes.data_as(_F), bb.ctypes.data_as(_F), n, out.ctypes.data_as(_F))
        assert rc == 0, oracle.lib.ezrt_last_error()
        return out
    return call


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _materials(rng, n):
    """Random Disney parameters, with the edge values the reference's branches test mixed in (metallic 1, roughness 0,
    clearcoatGloss 0 / 1 -> GTR1's a >= 1 branch is unreachable but a = 0.1 / 0.001 are, black base colour -> Cdlum = 0)."""
    m = rng.uniform(0.0, 1.0, (n, 18)).astype(np.float32)
    m[:, 0:3] = 0.0
    edge = rng.integers(0, 8, n)
    m[edge == 0, 7] = 1.0          # metallic
    m[edge == 1, 10] = 0.0         # roughness
    m[edge == 2, 15] = 1.0         # clearcoatGloss
    m[edge == 3, 15] = 0.0
    m[edge == 4, 3:6] = 0.0        # baseColor black: Ctint = vec3(1)
    m[edge == 5, 14] = 0.0         # clearcoat
    return m


N_FN = 100_000


def _vnl(rng, n):
    """V N L triples: N random, V and L mostly in N's hemisphere, a share below it (the early returns)."""
    N = _unit(rng, n)
    V, L = _unit(rng, n), _unit(rng, n)
    flip_v = (np.einsum("ij,ij->i", V, N) < 0) & (rng.uniform(size=n) < 0.85)
    flip_l = (np.einsum("ij,ij->i", L, N) < 0) & (rng.uniform(size=n) < 0.85)
    V[flip_v] = -V[flip_v]
    L[flip_l] = -L[flip_l]
    return np.concatenate([V, N, L], 1).astype(np.float32)


def test_fragment_seed_is_the_integer_pixel(fsh):
    """P5/fsh:315-318: uint((pix.x*0.5+0.5)*width) with pix = the pixel centre is the pixel index the oracle seeds with
    (SURVEY 8c "seed px,py = i,j"), for every column and row of the BASELINE frame sizes and some odd ones."""
    f = fsh[5]
    eye, cam = S.camera(0, 0, 4)
    for w, h in ((512, 512), (1024, 1024), (2048, 2048), (203, 117), (96, 80)):
        f.set_camera(eye, cam, w, h)
        for frame in (0, 5, 1000):
            xs = np.arange(w, dtype=np.uint64)
            got = np.array([f.seed(x, 3, frame) for x in range(w)], np.uint64)
            want = ((xs * 1973 + 3 * 9277 + frame * 26699) & 0xFFFFFFFF) | 1
            assert np.array_equal(got, want)
            got = np.array([f.seed(7, y, frame) for y in range(h)], np.uint64)
            want = ((7 * 1973 + np.arange(h, dtype=np.uint64) * 9277 + frame * 26699) & 0xFFFFFFFF) | 1
            assert np.array_equal(got, want)


def test_brdf_evaluate_chapter5_isotropic(fsh, ofn):
    rng = np.random.default_rng(1)
    a, m = _vnl(rng, N_FN), _materials(rng, N_FN)
    got, want = fsh[5].fn(1, a, m), ofn(None, 1, 5, a, m)
    assert same_bits(got, want) and float(np.abs(want).max()) > 0.1


def test_brdf_evaluate_of_the_uniform_loops(fsh, ofn):
    """Chapter 4's anisotropic BRDF_Evaluate (P4/fsh:412-473) and chapter 5's BRDF_Evaluate_aniso, whose body is the
    ISOTROPIC one (P5/fsh:465-471), each with X, Y from the shader's own getTangent."""
    rng = np.random.default_rng(2)
    a, m = _vnl(rng, N_FN), _materials(rng, N_FN)
    g4, w4 = fsh[4].fn(2, a, m), ofn(None, 2, 4, a, m)
    g5, w5 = fsh[5].fn(2, a, m), ofn(None, 2, 5, a, m)
    assert same_bits(g4, w4) and same_bits(g5, w5)
    assert not same_bits(g4, g5)                       # the two chapters do differ (anisotropic > 0)
    assert same_bits(g5, fsh[5].fn(1, a, m))           # Q12: chapter 5's "aniso" IS its isotropic evaluate


def test_sample_brdf_and_pdf(fsh, ofn):
    rng = np.random.default_rng(3)
    m = _materials(rng, N_FN)
    xi = rng.uniform(0, 1, (N_FN, 3)).astype(np.float32)
    xi[:64, 2] = np.float32(1.0)       # rand() can return exactly 1.0: the clearcoat branch's upper edge
    xi[64:128, 1] = np.float32(1.0)
    vn = _vnl(rng, N_FN)
    a = np.concatenate([xi, vn[:, 0:6]], 1)
    got, want = fsh[5].fn(3, a, m), ofn(None, 3, 5, a, m)
    assert same_bits(got, want)
    # the pdf of the directions just sampled (the use the integrator makes of it) and of random ones
    a2 = np.concatenate([vn[:, 0:6], want], 1)
    assert same_bits(fsh[5].fn(4, a2, m), ofn(None, 4, 5, a2, m))
    assert same_bits(fsh[5].fn(4, vn, m), ofn(None, 4, 5, vn, m))


def test_hemisphere_sampling(fsh, ofn):
    rng = np.random.default_rng(4)
    a = np.concatenate([rng.uniform(0, 1, (N_FN, 2)).astype(np.float32), _unit(rng, N_FN)], 1)
    a[:16, 0] = np.float32(1.0)
    a[16:32, 2:5] = np.float32([1, 0, 0])      # |N.x| > 0.999: the other helper axis
    assert same_bits(fsh[5].fn(9, a), ofn(None, 9, 5, a))


@pytest.mark.parametrize("bilinear", [1, 0])
def test_env_functions(fsh, ofn, oracle, bunny_small, bilinear):
    """SampleHdr, hdrPdf (sine of the ELEVATION, integer W*W/2: Q13), hdrColor on the scene's map + cache; chapter 3's
    clamp to 10 and chapter 4's unclamped sampleHdr with the NEAREST filter of those chapters."""
    rng = np.random.default_rng(5)
    so = oracle.scene_create(bunny_small.tri, bunny_small.nodes)
    so.set_env(bunny_small.hdr, bunny_small.cache, bilinear)
    for c in (3, 4, 5):
        fsh[c].set_env(bunny_small.hdr, bunny_small.cache, bilinear)
    L = _unit(rng, N_FN)
    L[:8] = np.float32([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 1e-20], [-1, 0, -1e-20]])
    L *= rng.uniform(0.5, 2.0, (N_FN, 1)).astype(np.float32)     # hdrColor normalises
    xi = rng.uniform(0, 1, (N_FN, 2)).astype(np.float32)
    xi[:4] = np.float32([[0, 0], [1, 1], [0, 1], [1, 0]])
    assert same_bits(fsh[5].fn(6, xi), ofn(so, 6, 5, xi))
    assert same_bits(fsh[5].fn(5, L), ofn(so, 5, 5, L))
    assert same_bits(fsh[5].fn(7, L), ofn(so, 7, 5, L))
    assert same_bits(fsh[4].fn(7, L), ofn(so, 7, 4, L))
    assert same_bits(fsh[3].fn(7, L), ofn(so, 7, 3, L))
    assert float(fsh[3].fn(7, L).max()) <= 10.0 < float(fsh[4].fn(7, L).max())


def _camera_rays(eye, cam, w, h, rng):
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    px = ((xs + 0.5) / w * 2 - 1 + rng.uniform(-0.5, 0.5, xs.shape) / w).ravel()
    py = ((ys + 0.5) / h * 2 - 1 + rng.uniform(-0.5, 0.5, xs.shape) / h).ravel()
    m = np.asarray(cam, np.float64).reshape(4, 4).T            # column-major -> rows
    d = px[:, None] * m[:3, 0] + py[:, None] * m[:3, 1] - 1.5 * m[:3, 2]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.broadcast_to(np.asarray(eye, np.float64), d.shape)
    return np.concatenate([o, d], 1).astype(np.float32)


@pytest.mark.parametrize("chapter", [3, 4, 5])
def test_hitbvh_whole_hit_record(fsh, ofn, oracle, bunny_small, chapter):
    """hitBVH (P5/fsh:254-306) with hitArray / hitTriangle / hitAABB under it: isHit, isInside, distance, hitPoint, the
    smooth normal (xy-projected barycentrics: +1e-7 in chapter 5, +-5e-5 in chapters 3/4 -- Q2) and the material of the
    winner, for camera rays, rays from surface points and rays with zero direction components (Q4: inf / NaN slabs)."""
    rng = np.random.default_rng(6)
    so = oracle.scene_create(bunny_small.tri, bunny_small.nodes)
    fsh[chapter].set_scene(bunny_small.tri, bunny_small.nodes)
    eye, cam = S.camera(10, 5, 3)
    rays = _camera_rays(eye, cam, 160, 120, rng)
    first = ofn(so, 8, chapter, rays)
    hit = first[:, 0] > 0
    assert 0.3 < hit.mean() < 0.99
    bounce = np.concatenate([first[hit, 3:6], _unit(rng, int(hit.sum()))], 1)
    axis = np.concatenate([rng.uniform(-1, 1, (600, 3)), np.tile(np.eye(3), (200, 1)) * rng.choice([-1, 1], (600, 1))], 1).astype(np.float32)
    allr = np.concatenate([rays, bounce, axis])
    got, want = fsh[chapter].fn(8, allr), ofn(so, 8, chapter, allr)
    assert same_bits(got, want)
    assert (want[:, 1] > 0).any()          # some rays hit from inside


FRAME_CASES = [  # (integrator, chapter, use_is, max_bounce, env filter, env clamp)
    (3, 3, 0, 2, 0, 10.0),
    (4, 4, 0, 4, 0, 0.0),
    (50, 5, 0, 4, 1, 0.0),
    (51, 5, 1, 2, 1, 0.0),
    (51, 5, 1, 3, 1, 0.0),
]


@pytest.mark.parametrize("integ,chapter,use_is,mb,bilinear,clamp", FRAME_CASES)
def test_main_per_pixel_sample_running_mean(fsh, oracle, bunny_small, integ, chapter, use_is, mb, bilinear, clamp):
    """The shader's main() -- seed, AA jitter, camera ray, hitBVH, the chapter's pathTracing loop, the lastFrame mix
    (P5/fsh:894-949) -- per pixel and frame, against ezrt_render of the oracle: the running mean after each of the frames
    0..5 of a 96x80 view, on the bits."""
    W, H = 96, 80
    eye, cam = S.camera(10, 5, 3)
    so = oracle.scene_create(bunny_small.tri, bunny_small.nodes)
    so.set_env(bunny_small.hdr, bunny_small.cache, bilinear)
    f = fsh[chapter]
    f.set_scene(bunny_small.tri, bunny_small.nodes)
    f.set_env(bunny_small.hdr, bunny_small.cache, bilinear)
    f.set_camera(eye, cam, W, H)
    f.set_integrator(mb, use_is)
    got = np.zeros((H, W, 4), np.float32)
    want = np.zeros((H, W, 4), np.float32)
    for frame in range(6):
        f.render(frame, 1, got)
        so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=1, frame0=frame, env_clamp=clamp), want)
        assert same_bits(got, want), "frame %d" % frame
    assert np.isfinite(want[..., :3]).mean() > 0.99 and float(np.nanmax(want[..., :3])) > 0.5
    # several frames in one call = the same recurrence
    again = f.render(0, 6)
    assert same_bits(again, want)
