/* ezrt_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic of EzRT's per-pixel trace, exposing
 * the C ABI of include/ezrt.h so parity tests are a two-library diff against
 * libezrt_hip.so.  Nothing in the product (ezrt_amd/) links, loads or calls
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do.
 *
 * PARITY PINNING STATUS: PINNED BY EXECUTION against the reference itself, compiled here from /root/reference by
 * oracle/ref_recipe/build_ref.py into oracle/_ref/ (nothing of it is copied into the repository):
 *   - host half (readObj, builders, encode, HDRLoader, calculateHdrCache, camera, the C++ twins of hitTriangle /
 *     hitAABB, chapter 2's probe ray): P2..P5 main.cpp + lib/hdrloader.cpp against GLM/GL stand-ins
 *     (tests/test_ref_pin.py, tests/test_gpu_ref_pin.py);
 *   - SHADER half (this file's restatement of P5/fsh:160-890, P4/fsh:412-517, P3/fsh:376-413): the three
 *     `shaders/fshader.fsh` themselves, compiled by g++ through a syntax-only source pass and a GLSL language shim
 *     (ref_recipe/fsh_pass.py, wrap_fsh.cpp, shim/glsl_shim.h): seed, BRDF_Evaluate (both chapters' forms),
 *     SampleBRDF, BRDF_Pdf, hdrPdf, SampleHdr, hdrColor, hemisphere sampling on 10^5 random inputs each, hitBVH's
 *     whole HitResult, and main()'s running mean per pixel and frame for the integrators 3 / 4 / 50 / 51 -- all
 *     bit-equal (tests/test_ref_fsh_pin.py); frames of the executed shaders are frozen in tests/golden/
 *     fsh_golden.npz and must be reproduced by this oracle AND by libezrt_hip.so (tests/test_fsh_golden.py);
 *   - the 30x3 Sobol known-answer table of T5 (tests/golden/sobol_kat.json).
 * What the reference leaves to the GLSL driver has nothing to be pinned against and is DEFINED here, shared by
 * oracle, shim and kernels: the precision of sin cos atan asin log pow (include/ezrt_detmath.h), the expansion of
 * dot / normalize / mix / reflect, min/max as (b<a)?b:a / (a<b)?b:a, texture filtering (texel centres, clamp to
 * edge, GL's bilinear formula in fp32), fp32 everywhere with no contraction.
 *
 * Each function cites the reference lines it follows.  Shorthands:
 *   P5/fsh = part 5 .../shaders/fshader.fsh   P4/fsh, P3/fsh likewise
 *   P2/main = part 2 .../main.cpp
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ezrt.h"
#include "ezrt_detmath.h"

#define PI EZ_PI
#define INF EZ_INF

/* ------------------------------------------------------------------------- */
/* vec3 with GLSL/GLM semantics fixed by SURVEY.md 2.3                        */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
  return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 vnormalize(v3 a) {
  float inv = 1.0f / __builtin_sqrtf(vdot(a, a));
  return vscale(a, inv);
}
static inline v3 vmix(v3 a, v3 b, float t) {
  return V3(ez_mix(a.x, b.x, t), ez_mix(a.y, b.y, t), ez_mix(a.z, b.z, t));
}
/* GLSL reflect(I, N) = I - 2.0 * dot(N, I) * N */
static inline v3 vreflect(v3 i, v3 n) {
  float k = 2.0f * vdot(n, i);
  return vsub(i, vscale(n, k));
}
static inline float sqr(float x) { return x * x; } /* P5/fsh:400 */

/* ------------------------------------------------------------------------- */
/* scene                                                                      */

typedef struct {
  v3 emissive, baseColor;
  float subsurface, metallic, specular, specularTint, roughness, anisotropic;
  float sheen, sheenTint, clearcoat, clearcoatGloss, IOR, transmission;
} Material;

typedef struct {
  int isHit, isInside;
  float distance;
  v3 hitPoint, normal, viewDir;
  int tri; /* winning triangle index (audit) */
  Material material;
} HitResult;

struct EzrtScene {
  int n_tri, n_nodes;
  float* tri;   /* n_tri * 36 */
  float* nodes; /* n_nodes * 12 */
  float* hdr;   /* w*h*3 or NULL */
  float* cache; /* w*h*3 or NULL */
  int env_w, env_h, env_filter;
  int instr;
  uint32_t sobol_mask; /* 7: dims wrap d & 7 (default), 15: sixteen dims (ezrt_scene_set_sampler) */
  uint64_t ctr[EZRT_CTR_COUNT];
  int64_t stats[6];
  float last_ms;
};

static __thread char g_err[256];
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}
const char* ezrt_last_error(void) { return g_err; }
int ezrt_trim(void) { return 0; } /* (nothing is cached between scenes on the CPU) */
const char* ezrt_backend(void) { return "oracle:cpu"; }

/* per-thread counters, merged after a parallel region */
typedef struct { uint64_t c[EZRT_CTR_COUNT]; } Ctr;

typedef struct {
  const struct EzrtScene* s;
  Ctr* ctr;
  int full; /* instrumentation level 1 */
  int p5tri; /* 1: P5 smooth-normal formula, 0: P3/P4 */
} Ctx;

/* getTriangle / getMaterial / getBVHNode: P5/fsh:93-155 */
static inline v3 tri_texel(const struct EzrtScene* s, int i, int k) {
  const float* p = s->tri + (size_t)i * 36 + k * 3;
  return V3(p[0], p[1], p[2]);
}
static Material get_material(const struct EzrtScene* s, int i) {
  Material m;
  v3 p1 = tri_texel(s, i, 8), p2 = tri_texel(s, i, 9), p3 = tri_texel(s, i, 10), p4 = tri_texel(s, i, 11);
  m.emissive = tri_texel(s, i, 6);
  m.baseColor = tri_texel(s, i, 7);
  m.subsurface = p1.x; m.metallic = p1.y; m.specular = p1.z;
  m.specularTint = p2.x; m.roughness = p2.y; m.anisotropic = p2.z;
  m.sheen = p3.x; m.sheenTint = p3.y; m.clearcoat = p3.z;
  m.clearcoatGloss = p4.x; m.IOR = p4.y; m.transmission = p4.z;
  return m;
}
typedef struct { int left, right, n, index; v3 AA, BB; } BVHNode;
static inline BVHNode get_node(const struct EzrtScene* s, int i) {
  const float* p = s->nodes + (size_t)i * 12;
  BVHNode n;
  n.left = (int)p[0]; n.right = (int)p[1]; /* ivec3(texelFetch) truncation, P5/fsh:143-148 */
  n.n = (int)p[3]; n.index = (int)p[4];
  n.AA = V3(p[6], p[7], p[8]);
  n.BB = V3(p[9], p[10], p[11]);
  return n;
}

/* ------------------------------------------------------------------------- */
/* hitTriangle: P5/fsh:160-217 (smooth normal +1e-7) and P3/fsh:228-282,
 * P4/fsh:151-204 (smooth normal +-5e-5); C++ twin P2/main.cpp:212-238.       */
static HitResult hit_triangle(const Ctx* cx, int i, v3 S, v3 d) {
  HitResult res;
  memset(&res, 0, sizeof res);
  res.distance = INF;
  res.tri = -1;
  const struct EzrtScene* s = cx->s;
  v3 p1 = tri_texel(s, i, 0), p2 = tri_texel(s, i, 1), p3 = tri_texel(s, i, 2);
  v3 N = vnormalize(vcross(vsub(p2, p1), vsub(p3, p1)));
  if (vdot(N, d) > 0.0f) {
    N = vneg(N);
    res.isInside = 1;
  }
  if (ez_abs(vdot(N, d)) < 0.00001f) return res;
  float t = (vdot(N, p1) - vdot(S, N)) / vdot(d, N);
  if (t < 0.0005f) return res;
  v3 P = vadd(S, vscale(d, t));
  v3 c1 = vcross(vsub(p2, p1), vsub(P, p1));
  v3 c2 = vcross(vsub(p3, p2), vsub(P, p2));
  v3 c3 = vcross(vsub(p1, p3), vsub(P, p3));
  float s1 = vdot(c1, N), s2 = vdot(c2, N), s3 = vdot(c3, N);
  int r1 = (s1 > 0.0f && s2 > 0.0f && s3 > 0.0f);
  int r2 = (s1 < 0.0f && s2 < 0.0f && s3 < 0.0f);
  if (r1 || r2) {
    res.isHit = 1;
    res.hitPoint = P;
    res.distance = t;
    res.viewDir = d;
    res.tri = i;
    v3 n1 = tri_texel(s, i, 3), n2 = tri_texel(s, i, 4), n3 = tri_texel(s, i, 5);
    float alpha, beta;
    if (cx->p5tri) { /* P5/fsh:206-207 */
      alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) /
              (-(p1.x - p2.x) * (p3.y - p2.y) + (p1.y - p2.y) * (p3.x - p2.x) + 1e-7f);
      beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) /
             (-(p2.x - p3.x) * (p1.y - p3.y) + (p2.y - p3.y) * (p1.x - p3.x) + 1e-7f);
    } else { /* P3/fsh:273-274, P4/fsh:196-197 */
      alpha = (-(P.x - p2.x) * (p3.y - p2.y) + (P.y - p2.y) * (p3.x - p2.x)) /
              (-(p1.x - p2.x - 0.00005f) * (p3.y - p2.y + 0.00005f) +
               (p1.y - p2.y + 0.00005f) * (p3.x - p2.x + 0.00005f));
      beta = (-(P.x - p3.x) * (p1.y - p3.y) + (P.y - p3.y) * (p1.x - p3.x)) /
             (-(p2.x - p3.x - 0.00005f) * (p1.y - p3.y + 0.00005f) +
              (p2.y - p3.y + 0.00005f) * (p1.x - p3.x + 0.00005f));
    }
    float gama = 1.0f - alpha - beta;
    v3 Ns = vadd(vadd(vscale(n1, alpha), vscale(n2, beta)), vscale(n3, gama));
    Ns = vnormalize(Ns);
    res.normal = res.isInside ? vneg(Ns) : Ns;
  }
  return res;
}

/* hitAABB: P5/fsh:220-233, P2/main.cpp:449-463 */
static float hit_aabb(v3 S, v3 d, v3 AA, v3 BB) {
  v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  v3 f = vmul(vsub(BB, S), inv);
  v3 n = vmul(vsub(AA, S), inv);
  v3 tmax = V3(ez_max(f.x, n.x), ez_max(f.y, n.y), ez_max(f.z, n.z));
  v3 tmin = V3(ez_min(f.x, n.x), ez_min(f.y, n.y), ez_min(f.z, n.z));
  float t1 = ez_min(tmax.x, ez_min(tmax.y, tmax.z));
  float t0 = ez_max(tmin.x, ez_max(tmin.y, tmin.z));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

/* hitArray: P5/fsh:238-251 */
static HitResult hit_array(const Ctx* cx, v3 S, v3 d, int l, int r) {
  HitResult res;
  memset(&res, 0, sizeof res);
  res.distance = INF;
  res.tri = -1;
  for (int i = l; i <= r; i++) {
    HitResult h = hit_triangle(cx, i, S, d);
    if (cx->full) cx->ctr->c[EZRT_CTR_TRI_TESTS]++;
    if (h.isHit && h.distance < res.distance) {
      res = h;
      res.material = get_material(cx->s, i);
      if (cx->full) cx->ctr->c[EZRT_CTR_MAT_FETCH]++;
    }
  }
  return res;
}

/* hitBVH: P5/fsh:254-306 (leaf range per P3-P5, not P2/main.cpp:471). */
static HitResult hit_bvh(const Ctx* cx, v3 S, v3 d) {
  HitResult res;
  memset(&res, 0, sizeof res);
  res.distance = INF;
  res.tri = -1;
  cx->ctr->c[EZRT_CTR_RAYS]++;
  int stack[256];
  int sp = 0;
  stack[sp++] = 1;
  while (sp > 0) {
    int top = stack[--sp];
    BVHNode node = get_node(cx->s, top);
    if (cx->full) cx->ctr->c[EZRT_CTR_NODE_POPS]++;
    if (node.n > 0) {
      int L = node.index, R = node.index + node.n - 1;
      HitResult r = hit_array(cx, S, d, L, R);
      if (r.isHit && r.distance < res.distance) res = r;
      continue;
    }
    if (cx->full) cx->ctr->c[EZRT_CTR_INNER_POPS]++;
    float d1 = INF, d2 = INF;
    if (node.left > 0) {
      BVHNode ln = get_node(cx->s, node.left);
      d1 = hit_aabb(S, d, ln.AA, ln.BB);
    }
    if (node.right > 0) {
      BVHNode rn = get_node(cx->s, node.right);
      d2 = hit_aabb(S, d, rn.AA, rn.BB);
    }
    if (d1 > 0.0f && d2 > 0.0f) {
      if (d1 < d2) {
        stack[sp++] = node.right;
        stack[sp++] = node.left;
      } else {
        stack[sp++] = node.left;
        stack[sp++] = node.right;
      }
    } else if (d1 > 0.0f) {
      stack[sp++] = node.left;
    } else if (d2 > 0.0f) {
      stack[sp++] = node.right;
    }
  }
  return res;
}

/* ------------------------------------------------------------------------- */
/* RNG: P5/fsh:315-331.  One state per pixel-sample.                          */

static inline uint32_t wang_hash(uint32_t* seed) {
  uint32_t s = *seed;
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  *seed = s;
  return s;
}
static inline float rnd(uint32_t* seed) { return (float)wang_hash(seed) / 4294967296.0f; }

/* Sobol: P5/fsh:351-376; dims 0-7 = the shader literal (8 dims x 32 bits); dims 8-15 (used only
 * after ezrt_scene_set_sampler(s, 16)) = the tutorial's recurrence (T5 tutorial.md:267-357) on the Joe-Kuo
 * parameters of dimensions 9-16, tools/gen_sobol_table.py --ext. */
static const uint32_t SOBOL_V[16 * 32] = {
#include "ezrt_sobol_v.inc"
#include "ezrt_sobol_v16.inc"
};
static inline uint32_t gray_code(uint32_t i) { return i ^ (i >> 1); }
static float sobol(uint32_t d, uint32_t i) {
  uint32_t result = 0, offset = d * 32u;
  for (uint32_t j = 0; i != 0; i >>= 1, j++)
    if (i & 1u) result ^= SOBOL_V[j + offset];
  return (float)result * (1.0f / (float)0xFFFFFFFFu);
}

/* CranleyPattersonRotation: P5/fsh:378-396 (114514/1919 = 59). */
static void cp_rotation(float* px, float* py, uint32_t ix, uint32_t iy) {
  uint32_t pseed = (ix * 1973u + iy * 9277u + 59u * 26699u) | 1u;
  float u = (float)wang_hash(&pseed) / 4294967296.0f;
  float v = (float)wang_hash(&pseed) / 4294967296.0f;
  float x = *px + u;
  if (x > 1.0f) x -= 1.0f;
  if (x < 0.0f) x += 1.0f;
  float y = *py + v;
  if (y > 1.0f) y -= 1.0f;
  if (y < 0.0f) y += 1.0f;
  *px = x;
  *py = y;
}

/* ------------------------------------------------------------------------- */
/* textures.  Definitions: texel-centre sampling, clamp-to-edge,
 * NEAREST = floor(u*W), BILINEAR = GL formula in fp32, lerp x then y.        */

static inline float sane01(float u) {
  if (!(u == u)) return 0.0f;
  return ez_clamp(u, 0.0f, 1.0f);
}
static v3 tex_fetch(const float* img, int W, int H, int filter, float u, float v) {
  u = sane01(u);
  v = sane01(v);
  if (filter == EZRT_FILTER_NEAREST) {
    int ix = (int)ez_floor(u * (float)W), iy = (int)ez_floor(v * (float)H);
    if (ix > W - 1) ix = W - 1;
    if (iy > H - 1) iy = H - 1;
    const float* p = img + ((size_t)iy * W + ix) * 3;
    return V3(p[0], p[1], p[2]);
  }
  float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
  float x0 = ez_floor(x), y0 = ez_floor(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
  if (ix0 < 0) ix0 = 0;
  if (iy0 < 0) iy0 = 0;
  if (ix1 > W - 1) ix1 = W - 1;
  if (iy1 > H - 1) iy1 = H - 1;
  const float* p00 = img + ((size_t)iy0 * W + ix0) * 3;
  const float* p10 = img + ((size_t)iy0 * W + ix1) * 3;
  const float* p01 = img + ((size_t)iy1 * W + ix0) * 3;
  const float* p11 = img + ((size_t)iy1 * W + ix1) * 3;
  v3 top = vmix(V3(p00[0], p00[1], p00[2]), V3(p10[0], p10[1], p10[2]), fx);
  v3 bot = vmix(V3(p01[0], p01[1], p01[2]), V3(p11[0], p11[1], p11[2]), fx);
  return vmix(top, bot, fy);
}

/* toSphericalCoord: P5/fsh:684-690 */
static void to_spherical(v3 v, float* pu, float* pv) {
  float u = ez_atan2(v.z, v.x), w = ez_asin(v.y);
  u = u / (2.0f * PI);
  w = w / PI;
  u = u + 0.5f;
  w = w + 0.5f;
  w = 1.0f - w;
  *pu = u;
  *pv = w;
}
/* hdrColor: P5/fsh:693-697; P3 clamp: P3/fsh:151-156 */
static v3 hdr_color(const Ctx* cx, v3 L, float env_clamp) {
  const struct EzrtScene* s = cx->s;
  if (cx->full) cx->ctr->c[EZRT_CTR_ENV_MAP]++;
  if (!s->hdr) return V3(0, 0, 0);
  float u, v;
  to_spherical(vnormalize(L), &u, &v);
  v3 c = tex_fetch(s->hdr, s->env_w, s->env_h, s->env_filter, u, v);
  if (env_clamp > 0.0f)
    c = V3(ez_min(c.x, env_clamp), ez_min(c.y, env_clamp), ez_min(c.z, env_clamp));
  return c;
}
/* SampleHdr: P5/fsh:667-679 */
static v3 sample_hdr(const Ctx* cx, float xi1, float xi2) {
  const struct EzrtScene* s = cx->s;
  if (cx->full) cx->ctr->c[EZRT_CTR_ENV_CACHE]++;
  v3 c = s->cache ? tex_fetch(s->cache, s->env_w, s->env_h, s->env_filter, xi1, xi2) : V3(0, 0, 0);
  float x = c.x, y = 1.0f - c.y;
  float phi = 2.0f * PI * (x - 0.5f);
  float theta = PI * (y - 0.5f);
  float st, ct, sp, cp;
  ez_sincos(theta, &st, &ct);
  ez_sincos(phi, &sp, &cp);
  return V3(ct * cp, st, ct * sp);
}
/* hdrPdf: P5/fsh:701-712 */
static float hdr_pdf(const Ctx* cx, v3 L) {
  const struct EzrtScene* s = cx->s;
  if (cx->full) cx->ctr->c[EZRT_CTR_ENV_CACHE]++;
  float u, v;
  to_spherical(vnormalize(L), &u, &v);
  float pdf = s->cache ? tex_fetch(s->cache, s->env_w, s->env_h, s->env_filter, u, v).z : 0.0f;
  float theta = PI * (0.5f - v);
  float sin_theta = ez_max(ez_sin(theta), 1e-10f);
  int res = s->env_w;
  float p_convert = (float)(res * res / 2) / (2.0f * PI * PI * sin_theta);
  return pdf * p_convert;
}

/* ------------------------------------------------------------------------- */
/* Disney principled BRDF: P5/fsh:400-549, P4/fsh:375-473                     */

static float schlick(float u) { /* P5/fsh:404-408 */
  float m = ez_clamp(1.0f - u, 0.0f, 1.0f);
  float m2 = m * m;
  return m2 * m2 * m;
}
static float gtr1(float NdotH, float a) { /* P5/fsh:410-415 */
  if (a >= 1.0f) return 1.0f / PI;
  float a2 = a * a;
  float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return (a2 - 1.0f) / (PI * ez_log(a2) * t);
}
static float gtr2(float NdotH, float a) { /* P5/fsh:417-421 */
  float a2 = a * a;
  float t = 1.0f + (a2 - 1.0f) * NdotH * NdotH;
  return a2 / (PI * t * t);
}
static float gtr2_aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) { /* :423 */
  return 1.0f / (PI * ax * ay * sqr(sqr(HdotX / ax) + sqr(HdotY / ay) + NdotH * NdotH));
}
static float smith_ggx(float NdotV, float alphaG) { /* P5/fsh:427-431 */
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return 1.0f / (NdotV + __builtin_sqrtf(a + b - a * b));
}
static float smith_ggx_aniso(float NdotV, float VdotX, float VdotY, float ax, float ay) { /* :433 */
  return 1.0f / (NdotV + __builtin_sqrtf(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
}

/* aniso = 0: P5 BRDF_Evaluate / BRDF_Evaluate_aniso (both isotropic bodies,
 * P5/fsh:437-549); aniso = 1: P4 BRDF_Evaluate (P4/fsh:412-473). */
static v3 brdf_evaluate(v3 V, v3 N, v3 L, v3 X, v3 Y, const Material* m, int aniso) {
  float NdotL = vdot(N, L), NdotV = vdot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return V3(0, 0, 0);
  v3 H = vnormalize(vadd(L, V));
  float NdotH = vdot(N, H), LdotH = vdot(L, H);

  v3 Cdlin = m->baseColor;
  float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
  v3 one = V3(1, 1, 1);
  v3 Ctint = (Cdlum > 0.0f) ? vdivs(Cdlin, Cdlum) : one;
  v3 Cspec = vscale(vmix(one, Ctint, m->specularTint), m->specular);
  v3 Cspec0 = vmix(vscale(Cspec, 0.08f), Cdlin, m->metallic);
  v3 Csheen = vmix(one, Ctint, m->sheenTint);

  float Fd90 = 0.5f + 2.0f * LdotH * LdotH * m->roughness;
  float FL = schlick(NdotL), FV = schlick(NdotV);
  float Fd = ez_mix(1.0f, Fd90, FL) * ez_mix(1.0f, Fd90, FV);

  float Fss90 = LdotH * LdotH * m->roughness;
  float Fss = ez_mix(1.0f, Fss90, FL) * ez_mix(1.0f, Fss90, FV);
  float ss = 1.25f * (Fss * (1.0f / (NdotL + NdotV) - 0.5f) + 0.5f);

  float Ds, Gs;
  float FH = schlick(LdotH);
  v3 Fs = vmix(Cspec0, one, FH);
  if (!aniso) {
    float alpha = ez_max(0.001f, sqr(m->roughness));
    Ds = gtr2(NdotH, alpha);
    Gs = smith_ggx(NdotL, m->roughness);
    Gs *= smith_ggx(NdotV, m->roughness);
  } else {
    float aspect = __builtin_sqrtf(1.0f - m->anisotropic * 0.9f);
    float ax = ez_max(0.001f, sqr(m->roughness) / aspect);
    float ay = ez_max(0.001f, sqr(m->roughness) * aspect);
    Ds = gtr2_aniso(NdotH, vdot(H, X), vdot(H, Y), ax, ay);
    Gs = smith_ggx_aniso(NdotL, vdot(L, X), vdot(L, Y), ax, ay);
    Gs *= smith_ggx_aniso(NdotV, vdot(V, X), vdot(V, Y), ax, ay);
  }

  float Dr = gtr1(NdotH, ez_mix(0.1f, 0.001f, m->clearcoatGloss));
  float Fr = ez_mix(0.04f, 1.0f, FH);
  float Gr = smith_ggx(NdotL, 0.25f) * smith_ggx(NdotV, 0.25f);

  v3 Fsheen = vscale(Csheen, FH * m->sheen);

  v3 diffuse = vadd(vscale(Cdlin, (1.0f / PI) * ez_mix(Fd, ss, m->subsurface)), Fsheen);
  v3 specular = vscale(vscale(Fs, Gs), Ds);
  float cc = 0.25f * Gr * Fr * Dr * m->clearcoat;
  v3 r = vadd(vscale(diffuse, 1.0f - m->metallic), specular);
  return vadd(r, V3(cc, cc, cc));
}

/* getTangent: P5/fsh:553-558 (names swapped in the reference: bitangent first) */
static void get_tangent(v3 N, v3* tangent, v3* bitangent) {
  v3 helper = V3(1, 0, 0);
  if (ez_abs(N.x) > 0.999f) helper = V3(0, 0, 1);
  *bitangent = vnormalize(vcross(N, helper));
  *tangent = vnormalize(vcross(N, *bitangent));
}
/* toNormalHemisphere: P5/fsh:561-567 */
static v3 to_normal_hemisphere(v3 v, v3 N) {
  v3 helper = V3(1, 0, 0);
  if (ez_abs(N.x) > 0.999f) helper = V3(0, 0, 1);
  v3 tangent = vnormalize(vcross(N, helper));
  v3 bitangent = vnormalize(vcross(N, tangent));
  return vadd(vadd(vscale(tangent, v.x), vscale(bitangent, v.y)), vscale(N, v.z));
}
/* SampleHemisphere: P5/fsh:570-576 (P3/fsh:109-114 with rand() arguments) */
static v3 sample_hemisphere(float xi1, float xi2) {
  float z = xi1;
  float r = ez_max(0.0f, __builtin_sqrtf(1.0f - z * z));
  float phi = 2.0f * PI * xi2;
  float s, c;
  ez_sincos(phi, &s, &c);
  return V3(r * c, r * s, z);
}
/* SampleCosineHemisphere: P5/fsh:579-590 */
static v3 sample_cosine_hemisphere(float xi1, float xi2, v3 N) {
  float r = __builtin_sqrtf(xi1);
  float theta = xi2 * 2.0f * PI;
  float s, c;
  ez_sincos(theta, &s, &c);
  float x = r * c, y = r * s;
  float z = __builtin_sqrtf(1.0f - x * x - y * y);
  return to_normal_hemisphere(V3(x, y, z), N);
}
/* SampleGTR2: P5/fsh:593-610 */
static v3 sample_gtr2(float xi1, float xi2, v3 V, v3 N, float alpha) {
  float phi_h = 2.0f * PI * xi1;
  float sin_phi_h, cos_phi_h;
  ez_sincos(phi_h, &sin_phi_h, &cos_phi_h);
  float cos_theta_h = __builtin_sqrtf((1.0f - xi2) / (1.0f + (alpha * alpha - 1.0f) * xi2));
  float sin_theta_h = __builtin_sqrtf(ez_max(0.0f, 1.0f - cos_theta_h * cos_theta_h));
  v3 H = V3(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
  H = to_normal_hemisphere(H, N);
  return vreflect(vneg(V), H);
}
/* SampleGTR1: P5/fsh:613-630 */
static v3 sample_gtr1(float xi1, float xi2, v3 V, v3 N, float alpha) {
  float phi_h = 2.0f * PI * xi1;
  float sin_phi_h, cos_phi_h;
  ez_sincos(phi_h, &sin_phi_h, &cos_phi_h);
  float cos_theta_h =
      __builtin_sqrtf((1.0f - ez_pow(alpha * alpha, 1.0f - xi2)) / (1.0f - alpha * alpha));
  float sin_theta_h = __builtin_sqrtf(ez_max(0.0f, 1.0f - cos_theta_h * cos_theta_h));
  v3 H = V3(sin_theta_h * cos_phi_h, sin_theta_h * sin_phi_h, cos_theta_h);
  H = to_normal_hemisphere(H, N);
  return vreflect(vneg(V), H);
}
/* SampleBRDF: P5/fsh:633-664 */
static v3 sample_brdf(float xi1, float xi2, float xi3, v3 V, v3 N, const Material* m) {
  float alpha_GTR1 = ez_mix(0.1f, 0.001f, m->clearcoatGloss);
  float alpha_GTR2 = ez_max(0.001f, sqr(m->roughness));
  float r_diffuse = 1.0f - m->metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m->clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float rd = xi3;
  if (rd <= p_diffuse) return sample_cosine_hemisphere(xi1, xi2, N);
  else if (p_diffuse < rd && rd <= p_diffuse + p_specular) return sample_gtr2(xi1, xi2, V, N, alpha_GTR2);
  else if (p_diffuse + p_specular < rd) return sample_gtr1(xi1, xi2, V, N, alpha_GTR1);
  return V3(0, 1, 0);
}
/* BRDF_Pdf: P5/fsh:715-752 */
static float brdf_pdf(v3 V, v3 N, v3 L, const Material* m) {
  float NdotL = vdot(N, L), NdotV = vdot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return 0.0f;
  v3 H = vnormalize(vadd(L, V));
  float NdotH = vdot(N, H);
  float LdotH = vdot(L, H);
  float alpha = ez_max(0.001f, sqr(m->roughness));
  float Ds = gtr2(NdotH, alpha);
  float Dr = gtr1(NdotH, ez_mix(0.1f, 0.001f, m->clearcoatGloss));
  float pdf_diffuse = NdotL / PI;
  float pdf_specular = Ds * NdotH / (4.0f * LdotH);
  float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
  float r_diffuse = 1.0f - m->metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m->clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float p_clearcoat = r_clearcoat / r_sum;
  float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
  return ez_max(1e-10f, pdf);
}
static float mis_mix_weight(float a, float b) { /* P5/fsh:754-757 */
  float t = a * a;
  return t / (b * b + t);
}

/* ---- importance sampling of the ANISOTROPIC specular lobe (integrator 52).
 * The reference stops short of it: chapter 4 evaluates GTR2_aniso but samples the hemisphere uniformly
 * (P4/fsh:412-473, 478-517); chapter 5 samples the isotropic lobe (SampleGTR2, P5/fsh:593-610) and keeps the
 * anisotropic evaluate as a commented block (P5/fsh:472-483).  Integrator 52 is chapter 5's loop
 * (P5/fsh:810-890) with that block switched on and the sampler / pdf that belong to it.  Nothing in the
 * reference pins these three functions, so the definitions below ARE the specification (evaluation order
 * as written, shared det-math): the GGX half-vector sampler of Burley 2012 ("Physically Based Shading at
 * Disney", B.2): h = normalize(sqrt(xi2 / (1 - xi2)) (ax cos(2 pi xi1) X + ay sin(2 pi xi1) Y) + N), whose
 * density is D(h) (N.h), i.e. pdf(L) = GTR2_aniso(h) N.h / (4 L.h) -- the form BRDF_Pdf uses for the
 * isotropic lobe (P5/fsh:731). */
static void aniso_alphas(const Material* m, float* ax, float* ay) { /* P4/fsh:441-443 = P5/fsh:474-476 */
  float aspect = __builtin_sqrtf(1.0f - m->anisotropic * 0.9f);
  *ax = ez_max(0.001f, sqr(m->roughness) / aspect);
  *ay = ez_max(0.001f, sqr(m->roughness) * aspect);
}
static v3 sample_gtr2_aniso(float xi1, float xi2, v3 V, v3 N, v3 X, v3 Y, float ax, float ay) {
  float phi_h = 2.0f * PI * xi1;
  float sin_phi_h, cos_phi_h;
  ez_sincos(phi_h, &sin_phi_h, &cos_phi_h);
  float k = __builtin_sqrtf(xi2 / ez_max(1e-7f, 1.0f - xi2)); /* tan(theta_h) of the unit-roughness lobe */
  v3 H = vadd(vadd(vscale(X, k * ax * cos_phi_h), vscale(Y, k * ay * sin_phi_h)), N);
  H = vnormalize(H);
  return vreflect(vneg(V), H);
}
/* SampleBRDF (P5/fsh:633-664) with the specular branch drawing from the anisotropic lobe */
static v3 sample_brdf_aniso(float xi1, float xi2, float xi3, v3 V, v3 N, v3 X, v3 Y, const Material* m) {
  float alpha_GTR1 = ez_mix(0.1f, 0.001f, m->clearcoatGloss);
  float ax, ay;
  aniso_alphas(m, &ax, &ay);
  float r_diffuse = 1.0f - m->metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m->clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float rd = xi3;
  if (rd <= p_diffuse) return sample_cosine_hemisphere(xi1, xi2, N);
  else if (p_diffuse < rd && rd <= p_diffuse + p_specular) return sample_gtr2_aniso(xi1, xi2, V, N, X, Y, ax, ay);
  else if (p_diffuse + p_specular < rd) return sample_gtr1(xi1, xi2, V, N, alpha_GTR1);
  return V3(0, 1, 0);
}
/* BRDF_Pdf (P5/fsh:715-752) with Ds = GTR2_aniso */
static float brdf_pdf_aniso(v3 V, v3 N, v3 L, v3 X, v3 Y, const Material* m) {
  float NdotL = vdot(N, L), NdotV = vdot(N, V);
  if (NdotL < 0.0f || NdotV < 0.0f) return 0.0f;
  v3 H = vnormalize(vadd(L, V));
  float NdotH = vdot(N, H);
  float LdotH = vdot(L, H);
  float ax, ay;
  aniso_alphas(m, &ax, &ay);
  float Ds = gtr2_aniso(NdotH, vdot(H, X), vdot(H, Y), ax, ay);
  float Dr = gtr1(NdotH, ez_mix(0.1f, 0.001f, m->clearcoatGloss));
  float pdf_diffuse = NdotL / PI;
  float pdf_specular = Ds * NdotH / (4.0f * LdotH);
  float pdf_clearcoat = Dr * NdotH / (4.0f * LdotH);
  float r_diffuse = 1.0f - m->metallic;
  float r_specular = 1.0f;
  float r_clearcoat = 0.25f * m->clearcoat;
  float r_sum = r_diffuse + r_specular + r_clearcoat;
  float p_diffuse = r_diffuse / r_sum;
  float p_specular = r_specular / r_sum;
  float p_clearcoat = r_clearcoat / r_sum;
  float pdf = p_diffuse * pdf_diffuse + p_specular * pdf_specular + p_clearcoat * pdf_clearcoat;
  return ez_max(1e-10f, pdf);
}

/* ------------------------------------------------------------------------- */
/* path logging for ezrt_render_paths                                         */
typedef struct { int32_t* tri; float* t; int n_slots; } PathLog;
static inline void plog(PathLog* pl, int slot, const HitResult* h) {
  if (!pl || slot >= pl->n_slots) return;
  pl->tri[slot] = h->isHit ? h->tri : -1;
  pl->t[slot] = h->isHit ? h->distance : INF;
}

typedef struct {
  const EzrtRenderParams* p;
  uint32_t frame;
  uint32_t ix, iy;
  uint32_t seed;
  uint32_t sobol_mask;
} Sample;

/* sobolVec2 + CP: P5/fsh:372-376, 845-846.  Dimensions beyond the 8 the table
 * holds wrap (d & 7) -- defined here; the reference indexes out of bounds.  With
 * ezrt_scene_set_sampler(s, 16) they wrap at 16 instead (8 bounces on distinct dimensions). */
static void sobol_cp(const Sample* sm, int bounce, float* u, float* v) {
  uint32_t g = gray_code(sm->frame + 1u);
  uint32_t d0 = ((uint32_t)bounce * 2u) & sm->sobol_mask, d1 = ((uint32_t)bounce * 2u + 1u) & sm->sobol_mask;
  *u = sobol(d0, g);
  *v = sobol(d1, g);
  cp_rotation(u, v, sm->ix, sm->iy);
}

/* pathTracing, three chapters: P3/fsh:376-413 (integrator 3), P4/fsh:478-517
 * (4), P5/fsh:762-807 (50). */
static v3 path_tracing_uniform(const Ctx* cx, Sample* sm, HitResult hit, PathLog* pl) {
  const EzrtRenderParams* p = sm->p;
  int integ = p->integrator;
  v3 Lo = V3(0, 0, 0), history = V3(1, 1, 1);
  for (int bounce = 0; bounce < p->max_bounce; bounce++) {
    v3 V = vneg(hit.viewDir), N = hit.normal;
    float xi1, xi2;
    if (integ == EZRT_INTEGRATOR_P5_SOBOL) {
      sobol_cp(sm, bounce, &xi1, &xi2);
    } else {
      xi1 = rnd(&sm->seed); /* z   (P3/fsh:110) */
      xi2 = rnd(&sm->seed); /* phi (P3/fsh:112) */
    }
    v3 L = to_normal_hemisphere(sample_hemisphere(xi1, xi2), N);
    float pdf = 1.0f / (2.0f * PI);
    float cosine_i = ez_max(0.0f, vdot(L, N));
    v3 f_r;
    if (integ == EZRT_INTEGRATOR_P3_DIFFUSE) {
      f_r = vdivs(hit.material.baseColor, PI);
    } else {
      v3 tangent, bitangent;
      get_tangent(N, &tangent, &bitangent);
      f_r = brdf_evaluate(V, N, L, tangent, bitangent, &hit.material, integ == EZRT_INTEGRATOR_P4_DISNEY);
    }
    HitResult nh = hit_bvh(cx, hit.hitPoint, L);
    plog(pl, 2 + 2 * bounce, &nh);
    if (!nh.isHit) {
      v3 sky = hdr_color(cx, L, p->env_clamp);
      Lo = vadd(Lo, vdivs(vscale(vmul(vmul(history, sky), f_r), cosine_i), pdf));
      break;
    }
    v3 Le = nh.material.emissive;
    Lo = vadd(Lo, vdivs(vscale(vmul(vmul(history, Le), f_r), cosine_i), pdf));
    hit = nh;
    history = vmul(history, vdivs(vscale(f_r, cosine_i), pdf));
  }
  return Lo;
}

/* pathTracingImportanceSampling: P5/fsh:810-890 (integrator 51); integrator 52 = the same loop with the
 * anisotropic specular lobe evaluated (the block P5/fsh:472-483 keeps commented out = P4/fsh:440-449),
 * sampled and priced (sample_brdf_aniso / brdf_pdf_aniso above). */
static v3 path_tracing_mis(const Ctx* cx, Sample* sm, HitResult hit, PathLog* pl) {
  const EzrtRenderParams* p = sm->p;
  const int aniso = p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO;
  v3 Lo = V3(0, 0, 0), history = V3(1, 1, 1);
  for (int bounce = 0; bounce < p->max_bounce; bounce++) {
    v3 V = vneg(hit.viewDir), N = hit.normal;
    v3 X = V3(0, 0, 0), Y = V3(0, 0, 0);
    if (aniso) get_tangent(N, &X, &Y); /* as chapter 4's loop does, P4/fsh:494-495 */
    float h1 = rnd(&sm->seed);
    float h2 = rnd(&sm->seed);
    v3 Lh = sample_hdr(cx, h1, h2);
    if (vdot(N, Lh) > 0.0f) {
      HitResult hh = hit_bvh(cx, hit.hitPoint, Lh);
      plog(pl, 1 + 2 * bounce, &hh);
      if (!hh.isHit) {
        v3 color = hdr_color(cx, Lh, p->env_clamp);
        float pdf_light = hdr_pdf(cx, Lh);
        v3 f_r = brdf_evaluate(V, N, Lh, X, Y, &hit.material, aniso);
        float pdf_brdf = aniso ? brdf_pdf_aniso(V, N, Lh, X, Y, &hit.material) : brdf_pdf(V, N, Lh, &hit.material);
        float w = mis_mix_weight(pdf_light, pdf_brdf);
        v3 c = vmul(vmul(vscale(history, w), color), f_r);
        Lo = vadd(Lo, vdivs(vscale(c, vdot(N, Lh)), pdf_light));
      }
    }
    float xi1, xi2;
    sobol_cp(sm, bounce, &xi1, &xi2);
    float xi3 = rnd(&sm->seed);
    v3 L = aniso ? sample_brdf_aniso(xi1, xi2, xi3, V, N, X, Y, &hit.material)
                 : sample_brdf(xi1, xi2, xi3, V, N, &hit.material);
    float NdotL = vdot(N, L);
    if (NdotL <= 0.0f) break;
    HitResult nh = hit_bvh(cx, hit.hitPoint, L);
    plog(pl, 2 + 2 * bounce, &nh);
    v3 f_r = brdf_evaluate(V, N, L, X, Y, &hit.material, aniso);
    float pdf_brdf = aniso ? brdf_pdf_aniso(V, N, L, X, Y, &hit.material) : brdf_pdf(V, N, L, &hit.material);
    if (pdf_brdf <= 0.0f) break;
    if (!nh.isHit) {
      v3 color = hdr_color(cx, L, p->env_clamp);
      float pdf_light = hdr_pdf(cx, L);
      float w = mis_mix_weight(pdf_brdf, pdf_light);
      v3 c = vmul(vmul(vscale(history, w), color), f_r);
      Lo = vadd(Lo, vdivs(vscale(c, NdotL), pdf_brdf));
      break;
    }
    v3 Le = nh.material.emissive;
    Lo = vadd(Lo, vdivs(vscale(vmul(vmul(history, Le), f_r), NdotL), pdf_brdf));
    hit = nh;
    history = vmul(history, vdivs(vscale(f_r, NdotL), pdf_brdf));
  }
  return Lo;
}

/* main(): P5/fsh:894-949 up to (not including) the lastFrame mix.            */
static v3 trace_sample(const Ctx* cx, const EzrtRenderParams* p, uint32_t ix, uint32_t iy, uint32_t frame,
                       PathLog* pl) {
  Sample sm;
  sm.p = p;
  sm.frame = frame;
  sm.ix = ix;
  sm.iy = iy;
  sm.seed = (ix * 1973u + iy * 9277u + frame * 26699u) | 1u; /* P5/fsh:315-318 */
  sm.sobol_mask = cx->s->sobol_mask;
  cx->ctr->c[EZRT_CTR_SAMPLES]++;
  float W = (float)p->width, H = (float)p->height;
  float pixx = ((float)ix + 0.5f) / W * 2.0f - 1.0f;
  float pixy = ((float)iy + 0.5f) / H * 2.0f - 1.0f;
  float aax = (rnd(&sm.seed) - 0.5f) / W;
  float aay = (rnd(&sm.seed) - 0.5f) / H;
  float vx = pixx + aax, vy = pixy + aay, vz = -1.5f;
  const float* m = p->camera_rotate;
  v3 dir = V3(m[0] * vx + m[4] * vy + m[8] * vz, m[1] * vx + m[5] * vy + m[9] * vz,
              m[2] * vx + m[6] * vy + m[10] * vz);
  dir = vnormalize(dir);
  v3 eye = V3(p->eye[0], p->eye[1], p->eye[2]);
  HitResult first = hit_bvh(cx, eye, dir);
  plog(pl, 0, &first);
  if (!first.isHit) return hdr_color(cx, dir, p->env_clamp);
  v3 Le = first.material.emissive;
  v3 Li = (p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO)
              ? path_tracing_mis(cx, &sm, first, pl)
                                                    : path_tracing_uniform(cx, &sm, first, pl);
  return vadd(Le, Li);
}

/* ------------------------------------------------------------------------- */
/* ABI                                                                        */

static int validate_params(const EzrtRenderParams* p) {
  if (!p) return fail(EZRT_ERR_INVALID, "params is NULL");
  if (p->width <= 0 || p->height <= 0) return fail(EZRT_ERR_INVALID, "width/height must be positive");
  if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 > p->x1 || p->y0 > p->y1)
    return fail(EZRT_ERR_INVALID, "pixel rect outside the image");
  if (p->max_bounce < 0 || p->max_bounce > 64) return fail(EZRT_ERR_INVALID, "max_bounce out of range [0,64]");
  if (p->integrator != 3 && p->integrator != 4 && p->integrator != 50 && p->integrator != 51 && p->integrator != 52)
    return fail(EZRT_ERR_INVALID, "unknown integrator");
  if (p->shard_count < 0 || p->shard_index < 0 || (p->shard_count > 0 && p->shard_index >= p->shard_count))
    return fail(EZRT_ERR_INVALID, "bad shard index/count");
  if (p->tile_w < 0 || p->tile_h < 0) return fail(EZRT_ERR_INVALID, "bad tile size");
  return 0;
}
static inline int pixel_owned(const EzrtRenderParams* p, int x, int y) {
  if (p->shard_count <= 1) return 1;
  int tw = p->tile_w > 0 ? p->tile_w : 32, th = p->tile_h > 0 ? p->tile_h : 32;
  int tiles_x = (p->width + tw - 1) / tw;
  int tile = (y / th) * tiles_x + (x / tw);
  return tile % p->shard_count == p->shard_index;
}

/* scene validation shared in spirit with the HIP library: node 0 is a dummy,
 * root is node 1, inner nodes have two children > 0, leaves index real
 * triangles, the tree is a tree (pre-order ids => child > parent). */
static int validate_scene(struct EzrtScene* s) {
  if (s->n_nodes < 2) return fail(EZRT_ERR_INVALID, "need at least the dummy node 0 and the root node 1");
  int64_t leaves = 0, maxleaf = 0;
  for (int i = 1; i < s->n_nodes; i++) {
    BVHNode n = get_node(s, i);
    if (n.n > 0) {
      if (n.index < 0 || (int64_t)n.index + n.n > s->n_tri)
        return fail(EZRT_ERR_INVALID, "leaf triangle range outside the triangle array");
      leaves++;
      if (n.n > maxleaf) maxleaf = n.n;
    } else {
      if (n.left <= i || n.right <= i || n.left >= s->n_nodes || n.right >= s->n_nodes)
        return fail(EZRT_ERR_INVALID, "inner node children must satisfy parent < child < nNodes");
    }
  }
  /* depth by forward propagation (children have larger ids) */
  int* depth = (int*)calloc((size_t)s->n_nodes, sizeof(int));
  if (!depth) return fail(EZRT_ERR_NOMEM, "out of memory");
  int maxd = 1;
  depth[1] = 1;
  for (int i = 1; i < s->n_nodes; i++) {
    BVHNode n = get_node(s, i);
    if (depth[i] == 0) continue; /* unreachable node */
    if (n.n <= 0) {
      depth[n.left] = depth[i] + 1;
      depth[n.right] = depth[i] + 1;
    }
    if (depth[i] > maxd) maxd = depth[i];
  }
  free(depth);
  if (maxd + 1 > 256) return fail(EZRT_ERR_UNSUPPORTED, "tree deeper than the reference's 256-entry stack");
  s->stats[0] = s->n_tri;
  s->stats[1] = s->n_nodes;
  s->stats[2] = maxd;
  s->stats[3] = leaves;
  s->stats[4] = maxleaf;
  s->stats[5] = (int64_t)s->n_tri * 144 + (int64_t)s->n_nodes * 48;
  return 0;
}

int ezrt_scene_create(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out) {
  if (!out) return fail(EZRT_ERR_INVALID, "out is NULL");
  *out = NULL;
  if (!tri || !nodes || n_tri <= 0 || n_nodes <= 0) return fail(EZRT_ERR_INVALID, "empty scene arrays");
  if (n_tri >= (1 << 24) || n_nodes >= (1 << 24))
    return fail(EZRT_ERR_UNSUPPORTED, "counts >= 2^24 are not exact in the float encoding");
  struct EzrtScene* s = (struct EzrtScene*)calloc(1, sizeof *s);
  if (!s) return fail(EZRT_ERR_NOMEM, "out of memory");
  s->n_tri = n_tri;
  s->n_nodes = n_nodes;
  s->sobol_mask = 7u;
  s->tri = (float*)malloc((size_t)n_tri * 36 * sizeof(float));
  s->nodes = (float*)malloc((size_t)n_nodes * 12 * sizeof(float));
  if (!s->tri || !s->nodes) {
    ezrt_scene_destroy(s);
    return fail(EZRT_ERR_NOMEM, "out of memory");
  }
  memcpy(s->tri, tri, (size_t)n_tri * 36 * sizeof(float));
  memcpy(s->nodes, nodes, (size_t)n_nodes * 12 * sizeof(float));
  int rc = validate_scene(s);
  if (rc) {
    ezrt_scene_destroy(s);
    return rc;
  }
  *out = s;
  return 0;
}
void ezrt_scene_destroy(EzrtScene* s) {
  if (!s) return;
  free(s->tri);
  free(s->nodes);
  free(s->hdr);
  free(s->cache);
  free(s);
}
int ezrt_scene_set_env(EzrtScene* s, const float* hdr, const float* cache, int w, int h, int filter) {
  if (!s || !hdr || w <= 0 || h <= 0) return fail(EZRT_ERR_INVALID, "bad env arguments");
  if (filter != EZRT_FILTER_NEAREST && filter != EZRT_FILTER_BILINEAR) return fail(EZRT_ERR_INVALID, "bad filter");
  if ((int64_t)w * w / 2 >= ((int64_t)1 << 31)) return fail(EZRT_ERR_UNSUPPORTED, "hdrResolution^2/2 overflows int");
  size_t n = (size_t)w * h * 3;
  free(s->hdr);
  free(s->cache);
  s->cache = NULL;
  s->hdr = (float*)malloc(n * sizeof(float));
  if (!s->hdr) return fail(EZRT_ERR_NOMEM, "out of memory");
  memcpy(s->hdr, hdr, n * sizeof(float));
  if (cache) {
    s->cache = (float*)malloc(n * sizeof(float));
    if (!s->cache) return fail(EZRT_ERR_NOMEM, "out of memory");
    memcpy(s->cache, cache, n * sizeof(float));
  }
  s->env_w = w;
  s->env_h = h;
  s->env_filter = filter;
  return 0;
}

static void merge_ctr(struct EzrtScene* s, const Ctr* c) {
  for (int k = 0; k < EZRT_CTR_COUNT; k++) s->ctr[k] += c->c[k];
}

#ifdef _OPENMP
#include <omp.h>
#endif
static double now_ms(void) {
#ifdef _OPENMP
  return omp_get_wtime() * 1e3;
#else
  return 0.0;
#endif
}

int ezrt_render_device(EzrtScene* s, const EzrtRenderParams* p, float* accum, void* stream) {
  (void)stream;
  if (!s || !accum) return fail(EZRT_ERR_INVALID, "scene/accum is NULL");
  int rc = validate_params(p);
  if (rc) return rc;
  if ((p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO) && !s->cache)
    return fail(EZRT_ERR_INVALID, "integrator 51 needs the env cache (ezrt_scene_set_env)");
  double t0 = now_ms();
  Ctr total;
  memset(&total, 0, sizeof total);
#pragma omp parallel
  {
    Ctr local;
    memset(&local, 0, sizeof local);
    Ctx cx;
    cx.s = s;
    cx.ctr = &local;
    cx.full = s->instr > 0;
    cx.p5tri = (p->integrator >= 50);
    /* tasks = 32-pixel pieces of a row, handed out dynamically: a 512-row image on a 256-core host would
     * otherwise give every thread two whole rows of very different cost */
    const int nxt = (p->x1 - p->x0 + 31) / 32, n_tasks = (p->y1 - p->y0) * (nxt > 0 ? nxt : 0);
#pragma omp for schedule(dynamic, 1)
    for (int task = 0; task < n_tasks; task++) {
      const int y = p->y0 + task / nxt, xs = p->x0 + (task % nxt) * 32, xe = xs + 32 < p->x1 ? xs + 32 : p->x1;
      for (int x = xs; x < xe; x++) {
        if (!pixel_owned(p, x, y)) continue;
        float* px = accum + ((size_t)y * p->width + x) * 4;
        v3 mean = V3(px[0], px[1], px[2]);
        for (uint32_t k = 0; k < p->spp; k++) {
          uint32_t frame = p->frame0 + k;
          v3 c = trace_sample(&cx, p, (uint32_t)x, (uint32_t)y, frame, NULL);
          /* mix(lastColor, color, 1/(frame+1)): P5/fsh:943-944.  Frame 0 has
           * weight 1: the previous content is ignored (defined, see ezrt.h). */
          if (frame == 0) {
            mean = c;
          } else {
            float a = 1.0f / (float)(frame + 1u);
            mean = vmix(mean, c, a);
          }
        }
        if (p->spp > 0) {
          px[0] = mean.x;
          px[1] = mean.y;
          px[2] = mean.z;
          px[3] = 1.0f;
        }
      }
    }
#pragma omp critical
    {
      for (int k = 0; k < EZRT_CTR_COUNT; k++) total.c[k] += local.c[k];
    }
  }
  merge_ctr(s, &total);
  s->last_ms = (float)(now_ms() - t0);
  return 0;
}
int ezrt_render(EzrtScene* s, const EzrtRenderParams* p, float* accum) {
  return ezrt_render_device(s, p, accum, NULL);
}

/* "device" frames are host memory here */
int ezrt_frame_create(int width, int height, float** frame_dev) {
  if (!frame_dev || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  *frame_dev = (float*)calloc((size_t)width * height * 4, sizeof(float));
  return *frame_dev ? 0 : fail(EZRT_ERR_NOMEM, "out of memory");
}
int ezrt_frame_destroy(float* frame_dev) {
  free(frame_dev);
  return 0;
}
int ezrt_frame_read(const float* frame_dev, int width, int height, float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  memcpy(rgba_host, frame_dev, (size_t)width * height * 16);
  return 0;
}
int ezrt_frame_write(float* frame_dev, int width, int height, const float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  memcpy(frame_dev, rgba_host, (size_t)width * height * 16);
  return 0;
}

int ezrt_frame_nonfinite(const float* frame_dev, int width, int height, void* stream, int64_t* n_pixels) {
  (void)stream;
  if (!frame_dev || !n_pixels || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  int64_t n = 0;
  for (size_t i = 0; i < (size_t)width * height; i++) { /* non-finite = exponent all ones */
    uint32_t u[3];
    memcpy(u, frame_dev + 4 * i, sizeof u);
    n += ((u[0] & 0x7f800000u) == 0x7f800000u) || ((u[1] & 0x7f800000u) == 0x7f800000u) || ((u[2] & 0x7f800000u) == 0x7f800000u);
  }
  *n_pixels = n;
  return 0;
}

int ezrt_render_paths(EzrtScene* s, const EzrtRenderParams* p, int32_t* tri_id, float* t_hit, float* colour) {
  if (!s || !tri_id || !t_hit) return fail(EZRT_ERR_INVALID, "NULL argument");
  int rc = validate_params(p);
  if (rc) return rc;
  if ((p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO) && !s->cache)
    return fail(EZRT_ERR_INVALID, "integrator 51 needs the env cache (ezrt_scene_set_env)");
  int slots = 1 + 2 * p->max_bounce;
  Ctr total;
  memset(&total, 0, sizeof total);
#pragma omp parallel
  {
    Ctr local;
    memset(&local, 0, sizeof local);
    Ctx cx;
    cx.s = s;
    cx.ctr = &local;
    cx.full = s->instr > 0;
    cx.p5tri = (p->integrator >= 50);
#pragma omp for schedule(dynamic, 1)
    for (int y = p->y0; y < p->y1; y++) {
      for (int x = p->x0; x < p->x1; x++) {
        if (!pixel_owned(p, x, y)) continue;
        size_t pix = (size_t)y * p->width + x;
        PathLog pl;
        pl.tri = tri_id + pix * slots;
        pl.t = t_hit + pix * slots;
        pl.n_slots = slots;
        for (int k = 0; k < slots; k++) {
          pl.tri[k] = -2;
          pl.t[k] = INF;
        }
        v3 c = trace_sample(&cx, p, (uint32_t)x, (uint32_t)y, p->frame0, &pl);
        if (colour) {
          colour[pix * 3 + 0] = c.x;
          colour[pix * 3 + 1] = c.y;
          colour[pix * 3 + 2] = c.z;
        }
      }
    }
#pragma omp critical
    {
      for (int k = 0; k < EZRT_CTR_COUNT; k++) total.c[k] += local.c[k];
    }
  }
  merge_ctr(s, &total);
  return 0;
}

int ezrt_query_hits(EzrtScene* s, const float* rays, int n_rays, int32_t* tri_id, float* t_hit) {
  if (!s || !rays || !tri_id || !t_hit || n_rays < 0) return fail(EZRT_ERR_INVALID, "NULL argument");
  Ctr total;
  memset(&total, 0, sizeof total);
#pragma omp parallel
  {
    Ctr local;
    memset(&local, 0, sizeof local);
    Ctx cx;
    cx.s = s;
    cx.ctr = &local;
    cx.full = s->instr > 0;
    cx.p5tri = 1;
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < n_rays; i++) {
      const float* r = rays + (size_t)i * 6;
      HitResult h = hit_bvh(&cx, V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]));
      tri_id[i] = h.isHit ? h.tri : -1;
      t_hit[i] = h.isHit ? h.distance : INF;
    }
#pragma omp critical
    {
      for (int k = 0; k < EZRT_CTR_COUNT; k++) total.c[k] += local.c[k];
    }
  }
  merge_ctr(s, &total);
  return 0;
}

/* pass3.fsh:14-24 + P1/main.cpp:187-189 */
int ezrt_tonemap(const float* rgba, int n_pixels, uint8_t* rgb8) {
  if (!rgba || !rgb8 || n_pixels < 0) return fail(EZRT_ERR_INVALID, "NULL argument");
  for (int i = 0; i < n_pixels; i++) {
    const float* c = rgba + (size_t)i * 4;
    float lum = 0.3f * c[0] + 0.6f * c[1] + 0.1f * c[2];
    float k = 1.0f + lum / 1.5f;
    for (int ch = 0; ch < 3; ch++) {
      float v = c[ch] * 1.0f / k; /* c * 1.0 / (1.0 + luminance / limit) */
      v = ez_pow(v, 1.0f / 2.2f);
      float q = ez_clamp(v * 255.0f, 0.0f, 255.0f);
      if (!(q == q)) q = 0.0f;
      rgb8[(size_t)i * 3 + ch] = (uint8_t)(int)q;
    }
  }
  return 0;
}

int ezrt_sobol(uint32_t index0, int n, int n_dims, float* out) {
  if (!out || n < 0 || n_dims < 1 || n_dims > 16) return fail(EZRT_ERR_INVALID, "bad sobol arguments");
  for (int i = 0; i < n; i++)
    for (int d = 0; d < n_dims; d++) out[(size_t)i * n_dims + d] = sobol((uint32_t)d, gray_code(index0 + (uint32_t)i));
  return 0;
}

int ezrt_scene_set_sampler(EzrtScene* s, int sobol_dims) {
  if (!s || (sobol_dims != 8 && sobol_dims != 16)) return fail(EZRT_ERR_INVALID, "sobol_dims must be 8 or 16");
  s->sobol_mask = (uint32_t)sobol_dims - 1u;
  return 0;
}

int ezrt_set_option(EzrtScene* s, const char* name, int value) {
  (void)value;
  if (!s || !name) return fail(EZRT_ERR_INVALID, "NULL argument");
  return 0; /* scheduling knobs of the GPU implementation: nothing to schedule here */
}

int ezrt_set_instrumentation(EzrtScene* s, int level) {
  if (!s || level < 0 || level > 1) return fail(EZRT_ERR_INVALID, "bad instrumentation level");
  s->instr = level;
  return 0;
}
int ezrt_counters(EzrtScene* s, uint64_t out[EZRT_CTR_COUNT]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  memcpy(out, s->ctr, sizeof s->ctr);
  return 0;
}
int ezrt_counters_reset(EzrtScene* s) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  memset(s->ctr, 0, sizeof s->ctr);
  return 0;
}
int ezrt_last_render_ms(EzrtScene* s, float* total_ms, float* trace_kernel_ms, int* n_trace_launches) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (total_ms) *total_ms = s->last_ms;
  if (trace_kernel_ms) *trace_kernel_ms = s->last_ms;
  if (n_trace_launches) *n_trace_launches = 1;
  return 0;
}
int ezrt_scene_prune_info(EzrtScene* s, double out[8]) { /* the oracle IS the unpruned traversal */
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  out[0] = -1.0;
  for (int k = 1; k < 8; k++) out[k] = 0.0;
  return 0;
}
int ezrt_scene_stats(EzrtScene* s, int64_t out[6]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  memcpy(out, s->stats, sizeof s->stats);
  return 0;
}

/* ezrt_oracle_fn -- ORACLE-ONLY export (not part of include/ezrt.h): function-level access to the restated shader
 * functions, so that tests/test_ref_fsh_pin.py can compare each of them with the REFERENCE'S OWN shader function compiled
 * by oracle/ref_recipe (wrap_fsh.cpp: fsh_fn, same op numbers and layouts).  p5: the chapter whose variant is meant
 * (5: P5 smooth-normal formula / isotropic evaluate; 4 and 3: the P3/P4 forms).
 *   1 BRDF_Evaluate (P5/fsh:500-549)            a = n x (V N L), b = n x material18       -> n x 3
 *   2 the uniform loop's evaluate with X, Y = getTangent(N): P4/fsh:412-473 (chapter 4, anisotropic) or
 *     P5/fsh:437-498 BRDF_Evaluate_aniso (chapter 5, isotropic body)                      -> n x 3
 *   3 SampleBRDF (P5/fsh:633-664)               a = n x (xi1 xi2 xi3 V N), b              -> n x 3
 *   4 BRDF_Pdf (P5/fsh:715-752)                 a = n x (V N L), b                        -> n
 *   5 hdrPdf (P5/fsh:701-712)                   a = n x L                                 -> n
 *   6 SampleHdr (P5/fsh:667-679)                a = n x (xi1 xi2)                         -> n x 3
 *   7 hdrColor / sampleHdr (P5/fsh:693-697; chapter 3 clamps to 10, P3/fsh:151-156)       -> n x 3
 *   8 hitBVH (P5/fsh:254-306)                   a = n x (S d) -> n x 12: isHit isInside distance hitPoint normal baseColor
 *   9 toNormalHemisphere(SampleHemisphere(xi1, xi2), N) (P5/fsh:561-576)  a = n x (xi1 xi2 N) -> n x 3 */
int ezrt_oracle_fn(EzrtScene* s, int op, int chapter, const float* a, const float* b, int n, float* out) {
  if (!a || !out || n < 0) return fail(EZRT_ERR_INVALID, "bad argument");
  Ctr local;
  memset(&local, 0, sizeof local);
  Ctx cx;
  cx.s = s;
  cx.ctr = &local;
  cx.full = 0;
  cx.p5tri = chapter >= 5;
  if ((op >= 5 && op <= 8) && !s) return fail(EZRT_ERR_INVALID, "this op needs a scene");
  for (int i = 0; i < n; i++) {
    Material m;
    memset(&m, 0, sizeof m);
    if (b && op >= 1 && op <= 4) {
      const float* q = b + (size_t)i * 18;
      m.emissive = V3(q[0], q[1], q[2]);
      m.baseColor = V3(q[3], q[4], q[5]);
      m.subsurface = q[6]; m.metallic = q[7]; m.specular = q[8]; m.specularTint = q[9]; m.roughness = q[10];
      m.anisotropic = q[11]; m.sheen = q[12]; m.sheenTint = q[13]; m.clearcoat = q[14]; m.clearcoatGloss = q[15];
      m.IOR = q[16]; m.transmission = q[17];
    }
    const float* p = a;
    v3 r = V3(0, 0, 0);
    switch (op) {
      case 1:
        p += (size_t)i * 9;
        r = brdf_evaluate(V3(p[0], p[1], p[2]), V3(p[3], p[4], p[5]), V3(p[6], p[7], p[8]), V3(0, 0, 0), V3(0, 0, 0), &m, 0);
        break;
      case 2: {
        p += (size_t)i * 9;
        v3 X, Y, N = V3(p[3], p[4], p[5]);
        get_tangent(N, &X, &Y);
        r = brdf_evaluate(V3(p[0], p[1], p[2]), N, V3(p[6], p[7], p[8]), X, Y, &m, chapter == 4);
        break;
      }
      case 3:
        p += (size_t)i * 9;
        r = sample_brdf(p[0], p[1], p[2], V3(p[3], p[4], p[5]), V3(p[6], p[7], p[8]), &m);
        break;
      case 4:
        p += (size_t)i * 9;
        out[i] = brdf_pdf(V3(p[0], p[1], p[2]), V3(p[3], p[4], p[5]), V3(p[6], p[7], p[8]), &m);
        continue;
      case 5:
        p += (size_t)i * 3;
        out[i] = hdr_pdf(&cx, V3(p[0], p[1], p[2]));
        continue;
      case 6:
        p += (size_t)i * 2;
        r = sample_hdr(&cx, p[0], p[1]);
        break;
      case 7:
        p += (size_t)i * 3;
        r = hdr_color(&cx, V3(p[0], p[1], p[2]), chapter == 3 ? 10.0f : 0.0f);
        break;
      case 8: {
        p += (size_t)i * 6;
        HitResult h = hit_bvh(&cx, V3(p[0], p[1], p[2]), V3(p[3], p[4], p[5]));
        float* o = out + (size_t)i * 12;
        for (int k = 0; k < 12; k++) o[k] = 0.0f;
        o[0] = h.isHit ? 1.0f : 0.0f;
        o[2] = h.distance;
        if (h.isHit) {
          o[1] = h.isInside ? 1.0f : 0.0f;
          o[3] = h.hitPoint.x; o[4] = h.hitPoint.y; o[5] = h.hitPoint.z;
          o[6] = h.normal.x; o[7] = h.normal.y; o[8] = h.normal.z;
          o[9] = h.material.baseColor.x; o[10] = h.material.baseColor.y; o[11] = h.material.baseColor.z;
        }
        continue;
      }
      case 9:
        p += (size_t)i * 5;
        r = to_normal_hemisphere(sample_hemisphere(p[0], p[1]), V3(p[2], p[3], p[4]));
        break;
      default: return fail(EZRT_ERR_INVALID, "unknown op");
    }
    out[(size_t)i * 3] = r.x;
    out[(size_t)i * 3 + 1] = r.y;
    out[(size_t)i * 3 + 2] = r.z;
  }
  return 0;
}

int ezrt_debug_math(int op, const float* a, const float* b, int n, float* out) {
  if (!a || !out || n < 0) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (op == 10 || op == 12) { /* hitAABB: a = n rays (S, d), b = n boxes (AA, BB) */
    if (!b) return fail(EZRT_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n; i++) {
      const float *r = a + 6 * (size_t)i, *q = b + 6 * (size_t)i;
      out[i] = hit_aabb(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]));
    }
    return 0;
  }
  if (op == 18) { /* the HIP library audits its fast correctly rounded reciprocal here; this library divides */
    for (int i = 0; i < n; i++) out[i] = 0.0f;
    return 0;
  }
  if (op == 17) { /* floor(bits(a[i]) / bits(b[0])): what the kernels' invariant-divisor form must return */
    uint32_t d = 0;
    if (!b) return fail(EZRT_ERR_INVALID, "NULL argument");
    memcpy(&d, b, 4);
    if (d == 0) return fail(EZRT_ERR_INVALID, "division by zero");
    for (int i = 0; i < n; i++) {
      uint32_t x, q;
      memcpy(&x, a + i, 4);
      q = x / d;
      memcpy(out + i, &q, 4);
    }
    return 0;
  }
  if (op >= 13 && op <= 16) { /* f4 audit in the frame N = (0,0,1), X/Y = getTangent(N).  b = n x (roughness,
                               * anisotropic, metallic, clearcoat, clearcoatGloss, -).  13: a = n x (V, L) ->
                               * brdf_pdf_aniso; 14/15/16: a = n x (xi1, xi2, xi3, V) -> sample_brdf_aniso .x/.y/.z */
    if (!b) return fail(EZRT_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n; i++) {
      const float *r = a + 6 * (size_t)i, *q = b + 6 * (size_t)i;
      Material m;
      memset(&m, 0, sizeof m);
      m.roughness = q[0];
      m.anisotropic = q[1];
      m.metallic = q[2];
      m.clearcoat = q[3];
      m.clearcoatGloss = q[4];
      v3 N = V3(0, 0, 1), X, Y;
      get_tangent(N, &X, &Y);
      if (op == 13) {
        out[i] = brdf_pdf_aniso(V3(r[0], r[1], r[2]), N, V3(r[3], r[4], r[5]), X, Y, &m);
      } else {
        v3 L = sample_brdf_aniso(r[0], r[1], r[2], V3(r[3], r[4], r[5]), N, X, Y, &m);
        out[i] = op == 14 ? L.x : (op == 15 ? L.y : L.z);
      }
    }
    return 0;
  }
  if (op == 11) { /* hitTriangle: a = n rays, b = n triangles (p1 p2 p3); out = t, INF on a miss */
    if (!b) return fail(EZRT_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n; i++) {
      float rec[36];
      memset(rec, 0, sizeof rec);
      memcpy(rec, b + 9 * (size_t)i, 9 * sizeof(float));
      struct EzrtScene tmp;
      memset(&tmp, 0, sizeof tmp);
      tmp.tri = rec;
      Ctr ctr;
      memset(&ctr, 0, sizeof ctr);
      Ctx cx = {&tmp, &ctr, 0, 1};
      const float* r = a + 6 * (size_t)i;
      HitResult h = hit_triangle(&cx, 0, V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]));
      out[i] = h.isHit ? h.distance : INF;
    }
    return 0;
  }
  for (int i = 0; i < n; i++) {
    float x = a[i], y = b ? b[i] : 0.0f, r;
    switch (op) {
      case 0: r = ez_sin(x); break;
      case 1: r = ez_cos(x); break;
      case 2: r = ez_atan2(x, y); break;
      case 3: r = ez_asin(x); break;
      case 4: r = ez_log(x); break;
      case 5: r = ez_exp(x); break;
      case 6: r = ez_pow(x, y); break;
      case 7: r = __builtin_sqrtf(x); break;
      case 8: r = x / y; break;
      case 9: { uint32_t sd = ez_f2u(x); r = rnd(&sd); break; }
      default: return fail(EZRT_ERR_INVALID, "unknown op");
    }
    out[i] = r;
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* include/ezrt_mgpu.h on the CPU: N shards of one frame rendered by the functions above, "transport" = memcpy.
 * Checks the sharding + pack / un-permute index rules (include/ezrt_tiles.h) without a GPU; device ordinals are
 * accepted and ignored. */
#include "ezrt_mgpu.h"
#include "ezrt_tiles.h"

struct EzrtMgpu {
  int n;
  EzrtScene* sc;       /* one scene serves every shard: host memory is shared */
  float** accum;       /* [n] full-size frame buffers, shard i valid in accum[i] */
  int width, height;
  EzrtRenderParams last;
  int have_last;
  float gather_ms;
  int64_t gather_bytes;
  float* render_ms;
};

int ezrt_mgpu_create(const float* tri, int n_tri, const float* nodes, int n_nodes, const int* devices, int n_devices,
                     int transport, EzrtMgpu** out) {
  if (!out) return fail(EZRT_ERR_INVALID, "out is NULL");
  *out = NULL;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(EZRT_ERR_INVALID, "need 1..64 devices");
  if (transport < 0 || transport > 2) return fail(EZRT_ERR_INVALID, "unknown transport");
  EzrtMgpu* m = (EzrtMgpu*)calloc(1, sizeof *m);
  if (!m) return fail(EZRT_ERR_NOMEM, "out of memory");
  m->n = n_devices;
  int rc = ezrt_scene_create(tri, n_tri, nodes, n_nodes, &m->sc);
  if (rc) {
    free(m);
    return rc;
  }
  m->accum = (float**)calloc((size_t)n_devices, sizeof(float*));
  m->render_ms = (float*)calloc((size_t)n_devices, sizeof(float));
  *out = m;
  return 0;
}
void ezrt_mgpu_destroy(EzrtMgpu* m) {
  if (!m) return;
  for (int i = 0; i < m->n; i++) free(m->accum[i]);
  free(m->accum);
  free(m->render_ms);
  ezrt_scene_destroy(m->sc);
  free(m);
}
int ezrt_mgpu_set_env(EzrtMgpu* m, const float* hdr, const float* cache, int w, int h, int filter) {
  if (!m) return fail(EZRT_ERR_INVALID, "NULL argument");
  return ezrt_scene_set_env(m->sc, hdr, cache, w, h, filter);
}
int ezrt_mgpu_set_sampler(EzrtMgpu* m, int sobol_dims) {
  if (!m) return fail(EZRT_ERR_INVALID, "NULL argument");
  return ezrt_scene_set_sampler(m->sc, sobol_dims);
}
int ezrt_mgpu_set_option(EzrtMgpu* m, const char* name, int value) {
  if (!m) return fail(EZRT_ERR_INVALID, "NULL argument");
  return ezrt_set_option(m->sc, name, value);
}
int ezrt_mgpu_render(EzrtMgpu* m, const EzrtRenderParams* p) {
  if (!m || !p) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (p->width <= 0 || p->height <= 0) return fail(EZRT_ERR_INVALID, "width/height must be positive");
  if (p->width != m->width || p->height != m->height) {
    for (int i = 0; i < m->n; i++) {
      free(m->accum[i]);
      m->accum[i] = (float*)calloc((size_t)p->width * p->height * 4, sizeof(float));
      if (!m->accum[i]) return fail(EZRT_ERR_NOMEM, "out of memory");
    }
    m->width = p->width;
    m->height = p->height;
  }
  for (int i = 0; i < m->n; i++) {
    EzrtRenderParams s = *p;
    s.shard_index = i;
    s.shard_count = m->n;
    int rc = ezrt_render_device(m->sc, &s, m->accum[i], NULL);
    if (rc) return rc;
    m->render_ms[i] = m->sc->last_ms;
  }
  m->last = *p;
  m->have_last = 1;
  return 0;
}
int ezrt_mgpu_gather(EzrtMgpu* m, float* accum_rgba) {
  if (!m) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (!m->have_last) return fail(EZRT_ERR_INVALID, "ezrt_mgpu_gather before any ezrt_mgpu_render");
  double t0 = now_ms();
  const EzrtTilePlan plan = ezrt_tile_plan(m->width, m->height, m->last.tile_w, m->last.tile_h, m->n);
  size_t total = 0;
  for (int i = 1; i < m->n; i++) {
    const size_t cnt = ezrt_tiles_packed_texels(&plan, i);
    float* packed = (float*)malloc((cnt ? cnt : 1) * 4 * sizeof(float));
    float* recv = (float*)malloc((cnt ? cnt : 1) * 4 * sizeof(float));
    if (!packed || !recv) {
      free(packed);
      free(recv);
      return fail(EZRT_ERR_NOMEM, "out of memory");
    }
    for (size_t k = 0; k < cnt; k++) { /* pack on "device" i */
      int x, y;
      if (ezrt_tiles_packed_to_pixel(&plan, i, k, &x, &y)) memcpy(packed + 4 * k, m->accum[i] + ((size_t)y * m->width + x) * 4, 16);
      else memset(packed + 4 * k, 0, 16);
    }
    memcpy(recv, packed, cnt * 16); /* the "transport" */
    for (size_t k = 0; k < cnt; k++) { /* un-permute on the root */
      int x, y;
      if (ezrt_tiles_packed_to_pixel(&plan, i, k, &x, &y)) memcpy(m->accum[0] + ((size_t)y * m->width + x) * 4, recv + 4 * k, 16);
    }
    free(packed);
    free(recv);
    total += cnt;
  }
  m->gather_ms = (float)(now_ms() - t0);
  m->gather_bytes = (int64_t)(total * 16);
  if (accum_rgba) memcpy(accum_rgba, m->accum[0], (size_t)m->width * m->height * 16);
  return 0;
}
int ezrt_mgpu_frame_device(EzrtMgpu* m, float** frame_dev) {
  if (!m || !frame_dev) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (!m->accum[0]) return fail(EZRT_ERR_INVALID, "no frame yet");
  *frame_dev = m->accum[0];
  return 0;
}
int ezrt_mgpu_counters(EzrtMgpu* m, uint64_t out[EZRT_CTR_COUNT]) {
  if (!m || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  return ezrt_counters(m->sc, out);
}
int ezrt_mgpu_last_ms(EzrtMgpu* m, float* render_ms, float* gather_ms, int64_t* gather_bytes) {
  if (!m) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (render_ms) memcpy(render_ms, m->render_ms, (size_t)m->n * sizeof(float));
  if (gather_ms) *gather_ms = m->gather_ms;
  if (gather_bytes) *gather_bytes = m->gather_bytes;
  return 0;
}
int64_t ezrt_tiles_packed_floats(int width, int height, int tile_w, int tile_h, int rank, int world) {
  if (width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  return (int64_t)(ezrt_tiles_packed_texels(&plan, rank) * 4);
}
int ezrt_tiles_pack_device(const float* accum_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                           float* packed_dev, void* stream) {
  (void)stream;
  if (!accum_dev || !packed_dev || width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world)
    return fail(EZRT_ERR_INVALID, "bad argument");
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  const size_t cnt = ezrt_tiles_packed_texels(&plan, rank);
  for (size_t k = 0; k < cnt; k++) {
    int x, y;
    if (ezrt_tiles_packed_to_pixel(&plan, rank, k, &x, &y)) memcpy(packed_dev + 4 * k, accum_dev + ((size_t)y * width + x) * 4, 16);
    else memset(packed_dev + 4 * k, 0, 16);
  }
  return 0;
}
int ezrt_tiles_unpack_device(const float* packed_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                             float* accum_dev, void* stream) {
  (void)stream;
  if (!accum_dev || !packed_dev || width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world)
    return fail(EZRT_ERR_INVALID, "bad argument");
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  const size_t cnt = ezrt_tiles_packed_texels(&plan, rank);
  for (size_t k = 0; k < cnt; k++) {
    int x, y;
    if (ezrt_tiles_packed_to_pixel(&plan, rank, k, &x, &y)) memcpy(accum_dev + ((size_t)y * width + x) * 4, packed_dev + 4 * k, 16);
  }
  return 0;
}
