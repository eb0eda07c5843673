// ezrt_device.h -- device-side building blocks of the gfx950 trace: vector
// helpers with a fixed evaluation order, RNG/Sobol, the BVH traversal over the
// device scene layout, texture fetches, the Disney BRDF and its samplers.
//
// Results contract: every function here reproduces, bit for bit, the fp32
// arithmetic the reference's fragment shader specifies once its
// implementation-defined parts are pinned as in DESIGN.md ("arithmetic
// contract"): left-to-right evaluation, no fma contraction (-ffp-contract=off),
// GLSL built-ins from include/ezrt_detmath.h.  Reference lines are cited per
// function (P5/fsh = part 5 shaders/fshader.fsh, etc.).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ezrt.h"
#include "ezrt_detmath.h"

#define EZD __device__ __forceinline__

namespace ezd {

constexpr float PI = EZ_PI;
constexpr float INF = EZ_INF;

struct f3 {
  float x, y, z;
};
EZD f3 mk(float x, float y, float z) { return f3{x, y, z}; }
// one pixel-sample's radiance as the shading stages hand it to accumulate_kernel: 12 B (the alpha of P5/fsh:946 is the
// constant 1: a float4 here was 25 % more traffic on both sides)
struct Sample3 {
  float x, y, z;
};
EZD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
EZD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
EZD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
EZD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
EZD f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
EZD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
EZD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
EZD f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// 1.0f / x, correctly rounded -- the same bits as the division the compiler emits for `1.0f / x` (IEEE binary32, round to nearest
// even: what gcc produces for the oracle), in 4 VALU instructions instead of the 10 + hazard nops of the general division macro
// (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup): v_rcp_f32 is within 1 ulp, one Newton step on the residual computed with
// a fused multiply-add (the rounding of THIS primitive's result is what the contract fixes, not how it is reached) lands on the
// correctly rounded quotient for every |x| in [2^-120, 2^120]; everything else -- zeros, denormals, huge values whose reciprocal is
// denormal, infinities, NaNs -- takes the compiler's division.  Not a claim: tests/test_gpu_parity.py runs ALL 2^32 bit patterns
// through both on the device (ezrt_debug_math op 18) and requires zero mismatches.  Rays per refill: three of these (1 / d) + the
// normalisation of a generated primary direction; the shading kernels' normalisations.
#ifndef EZRT_FAST_RCP
#define EZRT_FAST_RCP 1 // (0: A/B builds, tools/build_variant.sh)
#endif
EZD float ez_rcp(float x) {
  if (!EZRT_FAST_RCP) return 1.0f / x;
  const float ax = __builtin_fabsf(x);
  if (ax >= 7.5231638e-37f && ax <= 1.329228e36f) { // [2^-120, 2^120]
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(r, e, r);
    return r;
  }
  return 1.0f / x;
}
EZD f3 normalize(f3 a) {
  float inv = ez_rcp(__builtin_sqrtf(dot(a, a)));
  return a * inv;
}
EZD f3 mix3(f3 a, f3 b, float t) { return mk(ez_mix(a.x, b.x, t), ez_mix(a.y, b.y, t), ez_mix(a.z, b.z, t)); }
EZD f3 reflect(f3 i, f3 n) { // GLSL: I - 2.0 * dot(N, I) * N
  float k = 2.0f * dot(n, i);
  return i - n * k;
}
EZD float sqr(float x) { return x * x; }

// ---------------------------------------------------------------------------
// Device scene layout (built by ezrt_scene_create from the reference arrays).
//
//  tri_geom : 3 x float4 per triangle = 48 B, one cache line pair per test
//             (p1.xyz, N.x) (p2.xyz, N.y) (p3.xyz, N.z);  N = the unit plane
//             normal hitTriangle recomputes per call (P5/fsh:172) -- a pure
//             function of the triangle, evaluated once with the same fp32 ops.
//  tri_ref  : the reference's own 36-float records (144 B); only texels 3-11
//             (vertex normals + material) are read, once per *winning* hit.
//  inner    : 4 x float4 per inner node = 64 B: both children's boxes plus two
//             child references, so an inner visit is ONE 64-B fetch instead of
//             the reference's three 48-B getBVHNode calls (P5/fsh:266,281,285).
//             ref >= 0      : index of an inner record
//             ref bit31 set : leaf, bits 30..24 = n-1, bits 23..0 = first tri
struct DevScene {
  const float4* tri_geom;
  const float* tri_ref;
  const float4* inner;
  const float4* tri_shade; // 4 x float4 per triangle: what a WINNING hit needs besides tri_geom (see ShadeRec below)
  const float4* mat_table; // 7 x float4 per distinct material: its 18 floats + the constants brdf_evaluate derives from them
  uint32_t root_ref;
  int32_t n_tri;
  const float4* hdr;   // RGBA texels, row 0 = top
  const uint32_t* hdr_rgbe; // the same map as R | G << 8 | B << 16 | E << 24 when EVERY texel is exactly (m / 256) 2^(E - 128)
                            // per channel -- true by construction for maps HDRLoader decoded (hdrloader.cpp:97-114) -- else NULL
  const float4* cache; // (x/w, y/h, pdf, 0)
  const float2* cache_xy; // the same cache as two planes (or NULL): (x/w, y/h) for SampleHdr, pdf for hdrPdf -- the two
  const float* cache_pdf; // texels a bilinear lookup takes from a row are then ONE 16- or 8-byte load (tex_fetch_xy / _pdf)
  int32_t env_w, env_h, env_filter;
  uint32_t sobol_mask; // 7: Sobol dimensions wrap d & 7 (the reference's table), 15: sixteen dimensions (ezrt_scene_set_sampler)
};

// integrators 51 and 52 share pathTracingImportanceSampling's loop (P5/fsh:810-890); 52 swaps in the anisotropic lobe
template <int INTEG>
constexpr bool integ_mis() { return INTEG == EZRT_INTEGRATOR_P5_MIS || INTEG == EZRT_INTEGRATOR_P5_MIS_ANISO; }
template <int INTEG>
constexpr bool integ_aniso_is() { return INTEG == EZRT_INTEGRATOR_P5_MIS_ANISO; }

// Exact unsigned division by a launch-invariant divisor (Granlund-Montgomery, round-up form): with L = ceil(log2 d)
// and m = floor(2^32 (2^L - d) / d) + 1,  floor(n / d) = (((n - t) >> 1) + t) >> (L - 1),  t = mulhi(m, n),  for every
// 32-bit n.  The queue <-> sample-slot <-> pixel maps divide by the number of 16x16 blocks / of queue granules, values
// only known at launch: the compiler's 32-bit division is ~30 instructions, and ray generation runs it three times
// per sample (it was VALU-bound on them).  tests/test_host_scene.py checks the host twin over the divisors' range.
struct FastDiv {
  uint32_t d, m, s; // s = L - 1; d == 1: m = 0 and the quotient is n
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d ? d : 1u;
  f.m = 0u;
  f.s = 0u;
  if (f.d > 1u) {
    uint32_t L = 0;
    while ((1ull << L) < f.d) L++;
    f.m = (uint32_t)((((1ull << L) - f.d) << 32) / f.d) + 1u;
    f.s = L - 1u;
  }
  return f;
}
__host__ __device__ inline uint32_t fastdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1u) return n;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = __umulhi(f.m, n);
#else
  const uint32_t t = (uint32_t)(((unsigned long long)f.m * n) >> 32);
#endif
  return (((n - t) >> 1) + t) >> f.s;
}

constexpr uint32_t LEAF_BIT = 0x80000000u;

struct Counters { // per-thread, registers
  uint32_t rays, pops, inner, tris, mats, envmap, envcache;
};

// ---------------------------------------------------------------------------
// RNG: P5/fsh:315-331
// Work counters live in CTR_SLOTS copies (one 64-byte line each) and are summed by ezrt_counters: every wave adds
// its totals when it retires, and ONE counter word only sustains ~88 atomics/us chip-wide -- the 32 768 waves of
// the primary shading stage needed 370 us just to report their sample counts.
constexpr int CTR_SLOTS = 64;
EZD unsigned long long* ctr_slot(unsigned long long* base) {
  return base + (size_t)((blockIdx.x * 5u + (threadIdx.x >> 6)) & (CTR_SLOTS - 1)) * EZRT_CTR_COUNT;
}

// Lane mask of a predicate.  (HIP's __ballot takes an int: the bool is first materialised as 0/1
// in a VGPR and compared again -- two VALU slots per test in a VALU-bound loop.)
EZD unsigned long long ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

EZD uint32_t wang_hash(uint32_t& seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed *= 9u;
  seed = seed ^ (seed >> 4);
  seed *= 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
EZD float rnd(uint32_t& seed) { return (float)wang_hash(seed) / 4294967296.0f; }

// CranleyPattersonRotation: P5/fsh:378-396.  The offsets depend on the pixel only.
EZD void cp_offsets(uint32_t ix, uint32_t iy, float& u, float& v) {
  uint32_t pseed = (ix * 1973u + iy * 9277u + 59u * 26699u) | 1u;
  u = (float)wang_hash(pseed) / 4294967296.0f;
  v = (float)wang_hash(pseed) / 4294967296.0f;
}
EZD float cp_rotate(float p, float off) {
  p += off;
  if (p > 1.0f) p -= 1.0f;
  if (p < 0.0f) p += 1.0f;
  return p;
}

// ---------------------------------------------------------------------------
// hitAABB: P5/fsh:220-233 with invdir hoisted (1.0/dir is the same value on
// every call of one ray).
EZD float hit_aabb(f3 S, f3 inv, f3 AA, f3 BB) {
  f3 f = (BB - S) * inv;
  f3 n = (AA - S) * inv;
  float t1 = ez_min(ez_max(f.x, n.x), ez_min(ez_max(f.y, n.y), ez_max(f.z, n.z)));
  float t0 = ez_max(ez_min(f.x, n.x), ez_max(ez_min(f.y, n.y), ez_min(f.z, n.z)));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

// Same slab test on v_min_f32 / v_max_f32 / v_min3 / v_max3 (one issue slot each instead of
// compare + hazard nop + select).  Hardware min/max differ from (b<a)?b:a only on NaN operands and on
// the sign of a zero result; the result is consumed by comparisons only (sign of zero is invisible) and
// no NaN can arise when the ray's origin and 1/direction are finite (finite box minus finite origin
// times a finite factor is finite or +-inf, never NaN).  Callers use it only for such rays
// (ray_is_tame) and fall back to hit_aabb otherwise, so decisions stay bit-identical.
// (v_min/v_max straight from inline asm: through fminf/fmaxf the compiler first canonicalises every operand
// with a v_max_f32 x, x -- six more instructions per inner step of a loop that is bound by VALU issue)
EZD float hw_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
EZD float hw_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
EZD float hw_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
EZD float hw_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the two slab tests on boxes already translated by the ray origin (see TraceQArgs.inner_rel)
EZD float hit_aabb_rel(f3 inv, f3 AA, f3 BB) {
  f3 f = BB * inv;
  f3 n = AA * inv;
  float t1 = ez_min(ez_max(f.x, n.x), ez_min(ez_max(f.y, n.y), ez_max(f.z, n.z)));
  float t0 = ez_max(ez_min(f.x, n.x), ez_max(ez_min(f.y, n.y), ez_min(f.z, n.z)));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}
EZD float hit_aabb_tame_rel(f3 inv, f3 AA, f3 BB) {
  f3 f = BB * inv;
  f3 n = AA * inv;
  float t1 = hw_min3(hw_max(f.x, n.x), hw_max(f.y, n.y), hw_max(f.z, n.z));
  float t0 = hw_max3(hw_min(f.x, n.x), hw_min(f.y, n.y), hw_min(f.z, n.z));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}
EZD float hit_aabb_tame(f3 S, f3 inv, f3 AA, f3 BB) {
  f3 f = (BB - S) * inv;
  f3 n = (AA - S) * inv;
  float t1 = hw_min3(hw_max(f.x, n.x), hw_max(f.y, n.y), hw_max(f.z, n.z));
  float t0 = hw_max3(hw_min(f.x, n.x), hw_min(f.y, n.y), hw_min(f.z, n.z));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}
// hit_aabb_tame that also hands out the entry distance t0 = max_k of the near-plane products (for distance pruning)
EZD float hit_aabb_tame_e(f3 S, f3 inv, f3 AA, f3 BB, float& t0_out) {
  f3 f = (BB - S) * inv;
  f3 n = (AA - S) * inv;
  float t1 = hw_min3(hw_max(f.x, n.x), hw_max(f.y, n.y), hw_max(f.z, n.z));
  float t0 = hw_max3(hw_min(f.x, n.x), hw_min(f.y, n.y), hw_min(f.z, n.z));
  t0_out = t0;
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}
EZD bool ray_is_tame(f3 S, f3 inv) {
  const float big = 3.0e38f;
  return ez_abs(S.x) < big && ez_abs(S.y) < big && ez_abs(S.z) < big && ez_abs(inv.x) < big && ez_abs(inv.y) < big &&
         ez_abs(inv.z) < big;
}

// a finite ray with at least one direction component exactly +-0 (1/d = +-inf there): see ezrt_traceq4.h
EZD bool ray_is_semi(f3 S, f3 d, f3 inv) {
  const float big = 3.0e38f;
  const bool okx = ez_abs(inv.x) < big || d.x == 0.0f, oky = ez_abs(inv.y) < big || d.y == 0.0f, okz = ez_abs(inv.z) < big || d.z == 0.0f;
  return ez_abs(S.x) < big && ez_abs(S.y) < big && ez_abs(S.z) < big && okx && oky && okz && (d.x == 0.0f || d.y == 0.0f || d.z == 0.0f);
}

// hitTriangle, distance part: P5/fsh:160-198.  Flipping N (fsh:175-178) negates
// numerator, denominator and all three edge signs exactly, so t and the hit
// decision do not depend on it; isInside is recomputed for the winner.
EZD bool hit_triangle_t(const float4* __restrict__ g, f3 S, f3 d, float& t_out) {
  float4 a = g[0], b = g[1], c = g[2];
  f3 p1 = mk(a.x, a.y, a.z), p2 = mk(b.x, b.y, b.z), p3 = mk(c.x, c.y, c.z);
  f3 N = mk(a.w, b.w, c.w);
  float Nd = dot(N, d);
  if (ez_abs(Nd) < 0.00001f) return false;
  float t = (dot(N, p1) - dot(S, N)) / Nd;
  if (t < 0.0005f) return false;
  f3 P = S + d * t;
  f3 c1 = cross(p2 - p1, P - p1);
  f3 c2 = cross(p3 - p2, P - p2);
  f3 c3 = cross(p1 - p3, P - p3);
  float s1 = dot(c1, N), s2 = dot(c2, N), s3 = dot(c3, N);
  bool r1 = (s1 > 0.0f) && (s2 > 0.0f) && (s3 > 0.0f);
  bool r2 = (s1 < 0.0f) && (s2 < 0.0f) && (s3 < 0.0f);
  t_out = t;
  return r1 || r2;
}

// hitBVH: P5/fsh:254-306 + hitArray 238-251.  Unpruned, near-first, ties go
// right-first, strict < keeps the first-found hit -- identical visit order per
// ray.  The traversal stack lives in LDS: `stack` points at this lane's column
// (stride = STRIDE ints) so bank = lane % 32 and the two half-waves never
// conflict.  Only {t, triangle} are carried; everything else is a pure function
// of the winner.
template <bool FULLCTR, int STRIDE>
EZD void hit_bvh(const DevScene& sc, f3 S, f3 d, int* __restrict__ stack, int32_t& best_tri, float& best_t,
                 Counters& ctr) {
  ctr.rays++;
  best_tri = -1;
  best_t = INF;
  f3 inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
  int sp = 0;
  uint32_t ref = sc.root_ref;
  for (;;) {
    if (FULLCTR) ctr.pops++;
    if (ref & LEAF_BIT) {
      int first = (int)(ref & 0x00ffffffu);
      int n = (int)((ref >> 24) & 0x7fu) + 1;
      float leaf_best = INF; // only for the M counter (hitArray's local res)
      for (int i = first; i < first + n; i++) {
        float t;
        bool hit = hit_triangle_t(sc.tri_geom + (size_t)i * 3, S, d, t);
        if (FULLCTR) {
          ctr.tris++;
          if (hit && t < leaf_best) {
            leaf_best = t;
            ctr.mats++;
          }
        }
        if (hit && t < best_t) {
          best_t = t;
          best_tri = i;
        }
      }
    } else {
      if (FULLCTR) ctr.inner++;
      const float4* r = sc.inner + (size_t)ref * 4;
      float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
      float d1 = hit_aabb(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
      float d2 = hit_aabb(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
      uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
      if (d1 > 0.0f && d2 > 0.0f) {
        if (d1 < d2) { // left first: push right, continue with left
          stack[sp * STRIDE] = (int)right;
          sp++;
          ref = left;
        } else {
          stack[sp * STRIDE] = (int)left;
          sp++;
          ref = right;
        }
        continue;
      } else if (d1 > 0.0f) {
        ref = left;
        continue;
      } else if (d2 > 0.0f) {
        ref = right;
        continue;
      }
    }
    if (sp == 0) break;
    sp--;
    ref = (uint32_t)stack[sp * STRIDE];
  }
}

// ---------------------------------------------------------------------------
// Winner reconstruction: the rest of hitTriangle (P5/fsh:172-178, 199-214) and
// getMaterial (P5/fsh:110-135), evaluated once per ray for the closest hit.
struct Mat {
  f3 emissive, baseColor;
  float subsurface, metallic, specular, specularTint, roughness, anisotropic;
  float sheen, sheenTint, clearcoat, clearcoatGloss;
  // Sub-expressions of BRDF_Evaluate / SampleBRDF / BRDF_Pdf that depend on the material alone (P5/fsh:446-451, 468,
  // 486, 636-637, 727-728), evaluated ONCE per distinct material by mat_derive -- the same fp32 operations in the same
  // order, on the host at ezrt_scene_create (contraction off, ez_log from the shared header: bit-identical to the
  // device, tests/test_gpu_parity.py) -- instead of once per shaded hit: three divisions, a logarithm, seven mixes
  f3 Cspec0, Csheen;
  float alpha_gtr2;   // max(0.001, sqr(roughness))
  float alpha_gtr1;   // mix(0.1, 0.001, clearcoatGloss)
  float gtr1_a2m1;    // a2 - 1,         a2 = alpha_gtr1^2      (GTR1, P5/fsh:410-415)
  float gtr1_pilog;   // PI * log(a2)
};
struct Hit {
  f3 P, N, viewDir;
  Mat m;
};

EZD f3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

__host__ __device__ inline void mat_derive(Mat& m) {
  const float Cdlum = 0.3f * m.baseColor.x + 0.6f * m.baseColor.y + 0.1f * m.baseColor.z;
  const float tx = (Cdlum > 0.0f) ? m.baseColor.x / Cdlum : 1.0f, ty = (Cdlum > 0.0f) ? m.baseColor.y / Cdlum : 1.0f,
              tz = (Cdlum > 0.0f) ? m.baseColor.z / Cdlum : 1.0f;                                  // Ctint
  const float sx = ez_mix(1.0f, tx, m.specularTint) * m.specular, sy = ez_mix(1.0f, ty, m.specularTint) * m.specular,
              sz = ez_mix(1.0f, tz, m.specularTint) * m.specular;                                  // Cspec
  m.Cspec0 = f3{ez_mix(sx * 0.08f, m.baseColor.x, m.metallic), ez_mix(sy * 0.08f, m.baseColor.y, m.metallic),
                ez_mix(sz * 0.08f, m.baseColor.z, m.metallic)};
  m.Csheen = f3{ez_mix(1.0f, tx, m.sheenTint), ez_mix(1.0f, ty, m.sheenTint), ez_mix(1.0f, tz, m.sheenTint)};
  m.alpha_gtr2 = ez_max(0.001f, m.roughness * m.roughness);
  m.alpha_gtr1 = ez_mix(0.1f, 0.001f, m.clearcoatGloss);
  const float a2 = m.alpha_gtr1 * m.alpha_gtr1;
  m.gtr1_a2m1 = a2 - 1.0f;
  m.gtr1_pilog = EZ_PI * ez_log(a2);
}

// What a winning hit reads besides the 48-byte tri_geom record: 64 B per triangle, four aligned 16-byte loads
// (the reference record's texels 3-11 were seven, P5/fsh:110-135, 199-214):
//   (n1.xyz, n2.x) (n2.yz, n3.xy) (n3.z, bits(material index), -, -) (alpha and beta denominators of the smooth-normal
//   interpolation, P5/fsh:206-207 form and P3/fsh:273-274 form: functions of the triangle alone)
constexpr int SHADE_REC_FLOAT4 = 4, MAT_REC_FLOAT4 = 7;
struct ShadeDen { float a5, b5, a34, b34; };
__host__ __device__ inline ShadeDen shade_denominators(f3 p1, f3 p2, f3 p3) {
  ShadeDen d;
  d.a5 = -(p1.x - p2.x) * (p3.y - p2.y) + (p1.y - p2.y) * (p3.x - p2.x) + 1e-7f;
  d.b5 = -(p2.x - p3.x) * (p1.y - p3.y) + (p2.y - p3.y) * (p1.x - p3.x) + 1e-7f;
  d.a34 = -(p1.x - p2.x - 0.00005f) * (p3.y - p2.y + 0.00005f) + (p1.y - p2.y + 0.00005f) * (p3.x - p2.x + 0.00005f);
  d.b34 = -(p2.x - p3.x - 0.00005f) * (p1.y - p3.y + 0.00005f) + (p2.y - p3.y + 0.00005f) * (p1.x - p3.x + 0.00005f);
  return d;
}

template <bool P5TRI>
EZD void shade_point(const DevScene& sc, int32_t tri, float t, f3 S, f3 d, Hit& h) {
  const float4* g = sc.tri_geom + (size_t)tri * 3;
  float4 a = g[0], b = g[1], c = g[2];
  f3 p1 = mk(a.x, a.y, a.z), p2 = mk(b.x, b.y, b.z), p3 = mk(c.x, c.y, c.z);
  f3 N = mk(a.w, b.w, c.w);
  bool inside = dot(N, d) > 0.0f;
  f3 P = S + d * t;
  const float4* rq = sc.tri_shade + (size_t)tri * SHADE_REC_FLOAT4;
  const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2], r3 = rq[3];
  f3 n1 = mk(r0.x, r0.y, r0.z), n2 = mk(r0.w, r1.x, r1.y), n3 = mk(r1.z, r1.w, r2.x);
  const float4* mq = sc.mat_table + (size_t)__float_as_uint(r2.y) * MAT_REC_FLOAT4;
  const float4 m0 = mq[0], m1 = mq[1], m2 = mq[2], m3 = mq[3], m4 = mq[4], m5 = mq[5], m6 = mq[6];
  float alpha, beta;
  if (P5TRI) { // P5/fsh:206-207
    alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) / r3.x;
    beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) / r3.y;
  } else { // P3/fsh:273-274, P4/fsh:196-197
    alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) / r3.z;
    beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) / r3.w;
  }
  float gama = 1.0f - alpha - beta;
  f3 Ns = normalize(n1 * alpha + n2 * beta + n3 * gama);
  h.P = P;
  h.N = inside ? -Ns : Ns;
  h.viewDir = d;
  h.m.emissive = mk(m0.x, m0.y, m0.z);
  h.m.baseColor = mk(m0.w, m1.x, m1.y);
  h.m.subsurface = m1.z;
  h.m.metallic = m1.w;
  h.m.specular = m2.x;
  h.m.specularTint = m2.y;
  h.m.roughness = m2.z;
  h.m.anisotropic = m2.w;
  h.m.sheen = m3.x;
  h.m.sheenTint = m3.y;
  h.m.clearcoat = m3.z;
  h.m.clearcoatGloss = m3.w;
  // (m4.x, m4.y = IOR, transmission: carried by the reference, read by no shader)
  h.m.Cspec0 = mk(m4.z, m4.w, m5.x);
  h.m.Csheen = mk(m5.y, m5.z, m5.w);
  h.m.alpha_gtr2 = m6.x;
  h.m.alpha_gtr1 = m6.y;
  h.m.gtr1_a2m1 = m6.z;
  h.m.gtr1_pilog = m6.w;
}

// ---------------------------------------------------------------------------
// textures.  Defined (reference leaves it to the driver): texel-centre
// sampling, clamp-to-edge, NEAREST = floor(u*W), BILINEAR = GL formula in fp32
// with x-lerp then y-lerp.  NaN coordinates read texel 0.
EZD float sane01(float u) {
  if (!(u == u)) return 0.0f;
  return ez_clamp(u, 0.0f, 1.0f);
}
EZD f3 tex_fetch(const float4* __restrict__ img, int W, int H, int filter, float u, float v) {
  u = sane01(u);
  v = sane01(v);
  if (filter == EZRT_FILTER_NEAREST) {
    int ix = (int)ez_floor(u * (float)W), iy = (int)ez_floor(v * (float)H);
    if (ix > W - 1) ix = W - 1;
    if (iy > H - 1) iy = H - 1;
    float4 p = img[(size_t)iy * W + ix];
    return mk(p.x, p.y, p.z);
  }
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float x0 = ez_floor(x), y0 = ez_floor(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
  if (ix0 < 0) ix0 = 0;
  if (iy0 < 0) iy0 = 0;
  if (ix1 > W - 1) ix1 = W - 1;
  if (iy1 > H - 1) iy1 = H - 1;
  float4 p00 = img[(size_t)iy0 * W + ix0], p10 = img[(size_t)iy0 * W + ix1];
  float4 p01 = img[(size_t)iy1 * W + ix0], p11 = img[(size_t)iy1 * W + ix1];
  f3 top = mix3(mk(p00.x, p00.y, p00.z), mk(p10.x, p10.y, p10.z), fx);
  f3 bot = mix3(mk(p01.x, p01.y, p01.z), mk(p11.x, p11.y, p11.z), fx);
  return mix3(top, bot, fy);
}

// The environment map through its RGBE form when it has one (DevScene::hdr_rgbe): 4 bytes per texel instead of 16, so
// a 1024x512 map is 2 MB and stays in every XCD's L2 (the float4 map is 8 MB and the bounce rays' lookups are random).
// Decoding reproduces HDRLoader's convertComponent exactly: (m / 256) * 2^(E - 128) = ldexp(m, E - 136), exact in fp32
// down to the subnormals.
EZD f3 rgbe_texel(uint32_t p) {
  const int e = (int)(p >> 24) - 136;
  return mk(__builtin_amdgcn_ldexpf((float)(p & 255u), e), __builtin_amdgcn_ldexpf((float)((p >> 8) & 255u), e),
            __builtin_amdgcn_ldexpf((float)((p >> 16) & 255u), e));
}
EZD f3 tex_fetch_rgbe(const uint32_t* __restrict__ img, int W, int H, int filter, float u, float v) {
  u = sane01(u);
  v = sane01(v);
  if (filter == EZRT_FILTER_NEAREST) {
    int ix = (int)ez_floor(u * (float)W), iy = (int)ez_floor(v * (float)H);
    if (ix > W - 1) ix = W - 1;
    if (iy > H - 1) iy = H - 1;
    return rgbe_texel(img[(size_t)iy * W + ix]);
  }
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float x0 = ez_floor(x), y0 = ez_floor(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
  if (ix0 < 0) ix0 = 0;
  if (iy0 < 0) iy0 = 0;
  if (ix1 > W - 1) ix1 = W - 1;
  if (iy1 > H - 1) iy1 = H - 1;
  // The two texels of a row are neighbours (or, at the map's left / right edge, the same texel): ONE 8-byte load per row
  // from the pair (bx, bx + 1) that contains them -- the bounce rays' lookups are scattered, so the stage is bound by
  // the number of load requests, not by bytes.  (4-byte aligned: the struct says so.)
  struct __attribute__((packed, aligned(4))) TexelPair {
    uint32_t a, b;
  };
  uint32_t q00, q10, q01, q11;
  if (W >= 2) {
    int bx = ix0 < W - 2 ? ix0 : W - 2;
    const TexelPair r0 = *reinterpret_cast<const TexelPair*>(img + (size_t)iy0 * W + bx);
    const TexelPair r1 = *reinterpret_cast<const TexelPair*>(img + (size_t)iy1 * W + bx);
    q00 = ix0 == bx ? r0.a : r0.b;
    q10 = ix1 == bx ? r0.a : r0.b;
    q01 = ix0 == bx ? r1.a : r1.b;
    q11 = ix1 == bx ? r1.a : r1.b;
  } else {
    q00 = img[(size_t)iy0 * W + ix0], q10 = img[(size_t)iy0 * W + ix1];
    q01 = img[(size_t)iy1 * W + ix0], q11 = img[(size_t)iy1 * W + ix1];
  }
  f3 top = mix3(rgbe_texel(q00), rgbe_texel(q10), fx);
  f3 bot = mix3(rgbe_texel(q01), rgbe_texel(q11), fx);
  return mix3(top, bot, fy);
}

// Bilinear lookups in the two planes of the env cache (DevScene::cache_xy, cache_pdf): tex_fetch's arithmetic on the
// components the caller uses, with a row's two texels -- neighbours, or the same texel at the map's edge -- taken from
// the pair (bx, bx + 1) with one load.  W >= 2.
EZD void bilinear_taps(int W, int H, float u, float v, int& ix0, int& ix1, int& iy0, int& iy1, float& fx, float& fy) {
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float x0 = ez_floor(x), y0 = ez_floor(y);
  fx = x - x0;
  fy = y - y0;
  ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
  if (ix0 < 0) ix0 = 0;
  if (iy0 < 0) iy0 = 0;
  if (ix1 > W - 1) ix1 = W - 1;
  if (iy1 > H - 1) iy1 = H - 1;
}
EZD float tex_fetch_pdf(const float* __restrict__ img, int W, int H, float u, float v) {
  struct __attribute__((packed, aligned(4))) Pair {
    float a, b;
  };
  int ix0, ix1, iy0, iy1;
  float fx, fy;
  bilinear_taps(W, H, sane01(u), sane01(v), ix0, ix1, iy0, iy1, fx, fy);
  const int bx = ix0 < W - 2 ? ix0 : W - 2;
  const Pair r0 = *reinterpret_cast<const Pair*>(img + (size_t)iy0 * W + bx);
  const Pair r1 = *reinterpret_cast<const Pair*>(img + (size_t)iy1 * W + bx);
  const float p00 = ix0 == bx ? r0.a : r0.b, p10 = ix1 == bx ? r0.a : r0.b;
  const float p01 = ix0 == bx ? r1.a : r1.b, p11 = ix1 == bx ? r1.a : r1.b;
  return ez_mix(ez_mix(p00, p10, fx), ez_mix(p01, p11, fx), fy);
}
EZD void tex_fetch_xy(const float2* __restrict__ img, int W, int H, float u, float v, float& cx, float& cy) {
  struct __attribute__((aligned(8))) Pair {
    float2 a, b;
  };
  int ix0, ix1, iy0, iy1;
  float fx, fy;
  bilinear_taps(W, H, sane01(u), sane01(v), ix0, ix1, iy0, iy1, fx, fy);
  const int bx = ix0 < W - 2 ? ix0 : W - 2;
  const Pair r0 = *reinterpret_cast<const Pair*>(img + (size_t)iy0 * W + bx);
  const Pair r1 = *reinterpret_cast<const Pair*>(img + (size_t)iy1 * W + bx);
  const float2 p00 = ix0 == bx ? r0.a : r0.b, p10 = ix1 == bx ? r0.a : r0.b;
  const float2 p01 = ix0 == bx ? r1.a : r1.b, p11 = ix1 == bx ? r1.a : r1.b;
  cx = ez_mix(ez_mix(p00.x, p10.x, fx), ez_mix(p01.x, p11.x, fx), fy);
  cy = ez_mix(ez_mix(p00.y, p10.y, fx), ez_mix(p01.y, p11.y, fx), fy);
}

// toSphericalCoord: P5/fsh:684-690
EZD void to_spherical(f3 v, float& u, float& w) {
  u = ez_atan2(v.z, v.x);
  w = ez_asin(v.y);
  u = u / (2.0f * PI);
  w = w / PI;
  u = u + 0.5f;
  w = w + 0.5f;
  w = 1.0f - w;
}
// hdrColor: P5/fsh:693-697 (P3 clamp: P3/fsh:151-156)
template <bool FULLCTR>
EZD f3 hdr_color(const DevScene& sc, f3 L, float env_clamp, Counters& ctr) {
  if (FULLCTR) ctr.envmap++;
  if (!sc.hdr) return mk(0, 0, 0);
  float u, v;
  to_spherical(normalize(L), u, v);
  f3 c = sc.hdr_rgbe ? tex_fetch_rgbe(sc.hdr_rgbe, sc.env_w, sc.env_h, sc.env_filter, u, v)
                     : tex_fetch(sc.hdr, sc.env_w, sc.env_h, sc.env_filter, u, v);
  if (env_clamp > 0.0f) c = mk(ez_min(c.x, env_clamp), ez_min(c.y, env_clamp), ez_min(c.z, env_clamp));
  return c;
}
// SampleHdr: P5/fsh:667-679
template <bool FULLCTR>
EZD f3 sample_hdr(const DevScene& sc, float xi1, float xi2, Counters& ctr) {
  if (FULLCTR) ctr.envcache++;
  f3 c;
  if (sc.cache_xy && sc.env_filter == EZRT_FILTER_BILINEAR) {
    c.z = 0.0f;
    tex_fetch_xy(sc.cache_xy, sc.env_w, sc.env_h, xi1, xi2, c.x, c.y);
  } else {
    c = tex_fetch(sc.cache, sc.env_w, sc.env_h, sc.env_filter, xi1, xi2);
  }
  float x = c.x, y = 1.0f - c.y;
  float phi = 2.0f * PI * (x - 0.5f);
  float theta = PI * (y - 0.5f);
  float st, ct, sp, cp;
  ez_sincos(theta, &st, &ct);
  ez_sincos(phi, &sp, &cp);
  return mk(ct * cp, st, ct * sp);
}
// hdrPdf: P5/fsh:701-712
template <bool FULLCTR>
EZD float hdr_pdf(const DevScene& sc, f3 L, Counters& ctr) {
  if (FULLCTR) ctr.envcache++;
  float u, v;
  to_spherical(normalize(L), u, v);
  float pdf = (sc.cache_pdf && sc.env_filter == EZRT_FILTER_BILINEAR) ? tex_fetch_pdf(sc.cache_pdf, sc.env_w, sc.env_h, u, v)
                                                                     : tex_fetch(sc.cache, sc.env_w, sc.env_h, sc.env_filter, u, v).z;
  float theta = PI * (0.5f - v);
  float sin_theta = ez_max(ez_sin(theta), 1e-10f);
  int res = sc.env_w;
  float p_convert = (float)(res * res / 2) / (2.0f * PI * PI * sin_theta);
  return pdf * p_convert;
}

// hdrColor(L) and hdrPdf(L) of the SAME direction (P5/fsh:829-830, 873-874): both start with
// toSphericalCoord(normalize(L)) -- a software atan2 and asin, ~170 instructions -- evaluated once here; the same
// operations on the same operands, so the same bits as the two separate calls (the compiler does not merge them).
template <bool FULLCTR>
EZD void hdr_color_pdf(const DevScene& sc, f3 L, float env_clamp, Counters& ctr, f3& color, float& pdf_light) {
  if (FULLCTR) {
    ctr.envmap++;
    ctr.envcache++;
  }
  float u, v;
  to_spherical(normalize(L), u, v);
  if (!sc.hdr) {
    color = mk(0, 0, 0);
  } else {
    f3 c = sc.hdr_rgbe ? tex_fetch_rgbe(sc.hdr_rgbe, sc.env_w, sc.env_h, sc.env_filter, u, v)
                       : tex_fetch(sc.hdr, sc.env_w, sc.env_h, sc.env_filter, u, v);
    if (env_clamp > 0.0f) c = mk(ez_min(c.x, env_clamp), ez_min(c.y, env_clamp), ez_min(c.z, env_clamp));
    color = c;
  }
  float pdf = (sc.cache_pdf && sc.env_filter == EZRT_FILTER_BILINEAR) ? tex_fetch_pdf(sc.cache_pdf, sc.env_w, sc.env_h, u, v)
                                                                     : tex_fetch(sc.cache, sc.env_w, sc.env_h, sc.env_filter, u, v).z;
  float theta = PI * (0.5f - v);
  float sin_theta = ez_max(ez_sin(theta), 1e-10f);
  int res = sc.env_w;
  float p_convert = (float)(res * res / 2) / (2.0f * PI * PI * sin_theta);
  pdf_light = pdf * p_convert;
}

// ---------------------------------------------------------------------------
// Disney principled BRDF: P5/fsh:400-549 (isotropic), P4/fsh:375-473 (anisotropic)
EZD float schlick(float u) {
  float m = ez_clamp(1.0f - u, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
EZD float gtr1(float NdotH, float a) {
  if (a >= 1.0f) return 1.0f / PI;
  float a2 = a * a;
  float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return (a2 - 1.0f) / (PI * ez_log(a2) * t);
}
// the same with the material's precomputed a2 - 1 and PI * log(a2) (Mat)
EZD float gtr1_m(float NdotH, const Mat& m) {
  if (m.alpha_gtr1 >= 1.0f) return 1.0f / PI;
  float t = 1.0f + m.gtr1_a2m1 * NdotH * NdotH;
  return m.gtr1_a2m1 / (m.gtr1_pilog * t);
}
EZD float gtr2(float NdotH, float a) {
  float a2 = a * a;
  float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return a2 / (PI * t * t);
}
EZD float gtr2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) {
  return 1.0f / (PI * ax * ay * sqr(sqr(HdotX / ax) + sqr(HdotY / ay) + NdotH * NdotH));
}
EZD float smith_ggx(float NdotV, float alphaG) {
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return 1.0f / (NdotV + __builtin_sqrtf(a + b - a * b));
}
EZD float smith_ggx_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay) {
  return 1.0f / (NdotV + __builtin_sqrtf(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
}

template <bool ANISO>
EZD f3 brdf_evaluate(f3 V, f3 N, f3 L, f3 X, f3 Y, const Mat& m) {
  float NdotL = dot(N, L), NdotV = dot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return mk(0, 0, 0);
  f3 H = normalize(L + V);
  float NdotH = dot(N, H), LdotH = dot(L, H);

  // Cdlum, Ctint, Cspec, Cspec0, Csheen (P5/fsh:446-451): functions of the material alone -> Mat (mat_derive)
  const f3 Cdlin = m.baseColor, one = mk(1, 1, 1), Cspec0 = m.Cspec0, Csheen = m.Csheen;

  float Fd90 = 0.5f + 2.0f * LdotH * LdotH * m.roughness;
  float FL = schlick(NdotL), FV = schlick(NdotV);
  float Fd = ez_mix(1.0f, Fd90, FL) * ez_mix(1.0f, Fd90, FV);

  float Fss90 = LdotH * LdotH * m.roughness;
  float Fss = ez_mix(1.0f, Fss90, FL) * ez_mix(1.0f, Fss90, FV);
  float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);

  float Ds, Gs;
  float FH = schlick(LdotH);
  f3 Fs = mix3(Cspec0, one, FH);
  if (!ANISO) {
    Ds = gtr2(NdotH, m.alpha_gtr2);
    Gs = smith_ggx(NdotL, m.roughness);
    Gs *= smith_ggx(NdotV, m.roughness);
  } else {
    float aspect = __builtin_sqrtf(1.0f - m.anisotropic * 0.9f);
    float ax = ez_max(0.001f, sqr(m.roughness) / aspect);
    float ay = ez_max(0.001f, sqr(m.roughness) * aspect);
    Ds = gtr2_aniso(NdotH, dot(H, X), dot(H, Y), ax, ay);
    Gs = smith_ggx_aniso(NdotL, dot(L, X), dot(L, Y), ax, ay);
    Gs *= smith_ggx_aniso(NdotV, dot(V, X), dot(V, Y), ax, ay);
  }
  float Dr = gtr1_m(NdotH, m);
  float Fr = ez_mix(0.04f, 1.0f, FH);
  float Gr = smith_ggx(NdotL, 0.25f) * smith_ggx(NdotV, 0.25f);

  f3 Fsheen = Csheen * (FH * m.sheen);
  f3 diffuse = Cdlin * ((1.0f / PI) * ez_mix(Fd, ss, m.subsurface)) + Fsheen;
  f3 specular = (Fs * Gs) * Ds;
  float cc = 0.25f * Gr * Fr * Dr * m.clearcoat;
  return (diffuse * (1.0f - m.metallic) + specular) + mk(cc, cc, cc);
}

// getTangent: P5/fsh:553-558
EZD void get_tangent(f3 N, f3& tangent, f3& bitangent) {
  f3 helper = mk(1, 0, 0);
  if (ez_abs(N.x) > 0.999f) helper = mk(0, 0, 1);
  bitangent = normalize(cross(N, helper));
  tangent = normalize(cross(N, bitangent));
}
// toNormalHemisphere: P5/fsh:561-567
EZD f3 to_normal_hemisphere(f3 v, f3 N) {
  f3 helper = mk(1, 0, 0);
  if (ez_abs(N.x) > 0.999f) helper = mk(0, 0, 1);
  f3 tangent = normalize(cross(N, helper));
  f3 bitangent = normalize(cross(N, tangent));
  return (tangent * v.x + bitangent * v.y) + N * v.z;
}
// SampleHemisphere: P5/fsh:570-576
EZD f3 sample_hemisphere(float xi1, float xi2) {
  float z = xi1;
  float r = ez_max(0.0f, __builtin_sqrtf(1.0f - z * z));
  float phi = 2.0f * PI * xi2;
  float s, c;
  ez_sincos(phi, &s, &c);
  return mk(r * c, r * s, z);
}
// SampleCosineHemisphere: P5/fsh:579-590
EZD f3 sample_cosine_hemisphere(float xi1, float xi2, f3 N) {
  float r = __builtin_sqrtf(xi1);
  float theta = xi2 * 2.0f * PI;
  float s, c;
  ez_sincos(theta, &s, &c);
  float x = r * c, y = r * s;
  float z = __builtin_sqrtf(1.0f - x * x - y * y);
  return to_normal_hemisphere(mk(x, y, z), N);
}
// SampleGTR2 / SampleGTR1: P5/fsh:593-630
EZD f3 sample_gtr(float xi1, f3 V, f3 N, float cos_theta_h) {
  float phi_h = 2.0f * PI * xi1;
  float sin_phi_h, cos_phi_h;
  ez_sincos(phi_h, &sin_phi_h, &cos_phi_h);
  float sin_theta_h = __builtin_sqrtf(ez_max(0.0f, 1.0f - cos_theta_h * cos_theta_h));
  f3 H = mk(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
  H = to_normal_hemisphere(H, N);
  return reflect(-V, H);
}
// SampleBRDF: P5/fsh:633-664
EZD f3 sample_brdf(float xi1, float xi2, float xi3, f3 V, f3 N, const Mat& m) {
  const float alpha_GTR1 = m.alpha_gtr1, alpha_GTR2 = m.alpha_gtr2;
  float r_diffuse = 1.0f - m.metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m.clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float rd = xi3;
  if (rd <= p_diffuse) return sample_cosine_hemisphere(xi1, xi2, N);
  if (p_diffuse < rd && rd <= p_diffuse + p_specular) {
    float c = __builtin_sqrtf((1.0f - xi2) / (1.0f + (alpha_GTR2 * alpha_GTR2 - 1.0f) * xi2));
    return sample_gtr(xi1, V, N, c);
  }
  if (p_diffuse + p_specular < rd) {
    float c = __builtin_sqrtf((1.0f - ez_pow(alpha_GTR1 * alpha_GTR1, 1.0f - xi2)) / (1.0f - alpha_GTR1 * alpha_GTR1));
    return sample_gtr(xi1, V, N, c);
  }
  return mk(0, 1, 0);
}
// BRDF_Pdf: P5/fsh:715-752
EZD float brdf_pdf(f3 V, f3 N, f3 L, const Mat& m) {
  float NdotL = dot(N, L), NdotV = dot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return 0.0f;
  f3 H = normalize(L + V);
  float NdotH = dot(N, H), LdotH = dot(L, H);
  float Ds = gtr2(NdotH, m.alpha_gtr2);
  float Dr = gtr1_m(NdotH, m);
  float pdf_diffuse = NdotL / PI;
  float pdf_specular = Ds * NdotH / (4.0f * LdotH);
  float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
  float r_diffuse = 1.0f - m.metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m.clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float p_clearcoat = r_clearcoat / r_sum;
  float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
  return ez_max(1e-10f, pdf);
}
EZD float mis_mix_weight(float a, float b) { // P5/fsh:754-757
  float t = a * a;
  return t / (b * b + t);
}

// ---- SURVEY 8(f4), integrator 52: the anisotropic specular lobe (P4/fsh:440-449; commented out in P5/fsh:472-483)
// importance-sampled and priced.  Not in the reference: include/ezrt.h (EZRT_INTEGRATOR_P5_MIS_ANISO) names the
// specification (sample_gtr2_aniso, sample_brdf_aniso, brdf_pdf_aniso); these are the same operations in the same order.
EZD void aniso_alphas(const Mat& m, float& ax, float& ay) { // P4/fsh:441-443
  float aspect = __builtin_sqrtf(1.0f - m.anisotropic * 0.9f);
  ax = ez_max(0.001f, sqr(m.roughness) / aspect);
  ay = ez_max(0.001f, sqr(m.roughness) * aspect);
}
EZD f3 sample_gtr2_aniso(float xi1, float xi2, f3 V, f3 N, f3 X, f3 Y, float ax, float ay) {
  float phi_h = 2.0f * PI * xi1;
  float sin_phi_h, cos_phi_h;
  ez_sincos(phi_h, &sin_phi_h, &cos_phi_h);
  float k = __builtin_sqrtf(xi2 / ez_max(1e-7f, 1.0f - xi2));
  f3 H = (X * (k * ax * cos_phi_h) + Y * (k * ay * sin_phi_h)) + N;
  H = normalize(H);
  return reflect(-V, H);
}
EZD f3 sample_brdf_aniso(float xi1, float xi2, float xi3, f3 V, f3 N, f3 X, f3 Y, const Mat& m) {
  const float alpha_GTR1 = m.alpha_gtr1;
  float ax, ay;
  aniso_alphas(m, ax, ay);
  float r_diffuse = 1.0f - m.metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m.clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float rd = xi3;
  if (rd <= p_diffuse) return sample_cosine_hemisphere(xi1, xi2, N);
  if (p_diffuse < rd && rd <= p_diffuse + p_specular) return sample_gtr2_aniso(xi1, xi2, V, N, X, Y, ax, ay);
  if (p_diffuse + p_specular < rd) {
    float c = __builtin_sqrtf((1.0f - ez_pow(alpha_GTR1 * alpha_GTR1, 1.0f - xi2)) / (1.0f - alpha_GTR1 * alpha_GTR1));
    return sample_gtr(xi1, V, N, c);
  }
  return mk(0, 1, 0);
}
EZD float brdf_pdf_aniso(f3 V, f3 N, f3 L, f3 X, f3 Y, const Mat& m) {
  float NdotL = dot(N, L), NdotV = dot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return 0.0f;
  f3 H = normalize(L + V);
  float NdotH = dot(N, H), LdotH = dot(L, H);
  float ax, ay;
  aniso_alphas(m, ax, ay);
  float Ds = gtr2_aniso(NdotH, dot(H, X), dot(H, Y), ax, ay);
  float Dr = gtr1_m(NdotH, m);
  float pdf_diffuse = NdotL / PI;
  float pdf_specular = Ds * NdotH / (4.0f * LdotH);
  float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
  float r_diffuse = 1.0f - m.metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m.clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float p_clearcoat = r_clearcoat / r_sum;
  float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
  return ez_max(1e-10f, pdf);
}

// BRDF_Evaluate(V, N, L) and BRDF_Pdf(V, N, L) of the SAME direction (P5/fsh:832-833, 858-859): both start with
// H = normalize(L + V), N.H, L.H and both evaluate GTR2 / GTR1 of that half vector -- once here.  The same operations
// on the same operands as brdf_evaluate<ANISO> followed by brdf_pdf (ANISO: brdf_pdf_aniso), hence the same bits; the
// compiler does not merge the two calls (a normalisation, two divisions and the GTR terms per pair).
template <bool ANISO>
EZD void brdf_evaluate_pdf(f3 V, f3 N, f3 L, f3 X, f3 Y, const Mat& m, f3& f_r, float& pdf_out) {
  float NdotL = dot(N, L), NdotV = dot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) {
    f_r = mk(0, 0, 0);
    pdf_out = 0.0f;
    return;
  }
  f3 H = normalize(L + V);
  float NdotH = dot(N, H), LdotH = dot(L, H);
  const f3 Cdlin = m.baseColor, one = mk(1, 1, 1), Cspec0 = m.Cspec0, Csheen = m.Csheen;

  float Fd90 = 0.5f + 2.0f * LdotH * LdotH * m.roughness;
  float FL = schlick(NdotL), FV = schlick(NdotV);
  float Fd = ez_mix(1.0f, Fd90, FL) * ez_mix(1.0f, Fd90, FV);

  float Fss90 = LdotH * LdotH * m.roughness;
  float Fss = ez_mix(1.0f, Fss90, FL) * ez_mix(1.0f, Fss90, FV);
  float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);

  float Ds, Gs;
  float FH = schlick(LdotH);
  f3 Fs = mix3(Cspec0, one, FH);
  if (!ANISO) {
    Ds = gtr2(NdotH, m.alpha_gtr2);
    Gs = smith_ggx(NdotL, m.roughness);
    Gs *= smith_ggx(NdotV, m.roughness);
  } else {
    float aspect = __builtin_sqrtf(1.0f - m.anisotropic * 0.9f);
    float ax = ez_max(0.001f, sqr(m.roughness) / aspect);
    float ay = ez_max(0.001f, sqr(m.roughness) * aspect);
    Ds = gtr2_aniso(NdotH, dot(H, X), dot(H, Y), ax, ay);
    Gs = smith_ggx_aniso(NdotL, dot(L, X), dot(L, Y), ax, ay);
    Gs *= smith_ggx_aniso(NdotV, dot(V, X), dot(V, Y), ax, ay);
  }
  float Dr = gtr1_m(NdotH, m);
  float Fr = ez_mix(0.04f, 1.0f, FH);
  float Gr = smith_ggx(NdotL, 0.25f) * smith_ggx(NdotV, 0.25f);

  f3 Fsheen = Csheen * (FH * m.sheen);
  f3 diffuse = Cdlin * ((1.0f / PI) * ez_mix(Fd, ss, m.subsurface)) + Fsheen;
  f3 specular = (Fs * Gs) * Ds;
  float cc = 0.25f * Gr * Fr * Dr * m.clearcoat;
  f_r = (diffuse * (1.0f - m.metallic) + specular) + mk(cc, cc, cc);

  // BRDF_Pdf: P5/fsh:729-751 on the same H, Ds, Dr
  float pdf_diffuse = NdotL / PI;
  float pdf_specular = Ds * NdotH / (4.0f * LdotH);
  float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
  float r_diffuse = 1.0f - m.metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m.clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float p_clearcoat = r_clearcoat / r_sum;
  float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
  pdf_out = ez_max(1e-10f, pdf);
}

} // namespace ezd
