/* ezrt_detmath.h -- deterministic binary32 definitions of the GLSL / libm
 * built-ins the EzRT trace path uses.
 *
 * Why this exists: the reference evaluates sin/cos/atan/asin/pow/log inside a
 * GLSL fragment shader (P5/shaders/fshader.fsh:574-575,596-597,619,672-676,685,
 * 414) whose precision is implementation-defined ("parity
 * unpinned").  The radiance integrand is discontinuous (hit/miss, lobe choice),
 * so the 1e-4 L-inf bar is only reachable if the CPU oracle and the gfx950
 * kernels make bit-identical decisions.  These functions are therefore
 * *definitions*: built only from + - * / sqrt, float<->int conversion and bit
 * casts, which are correctly rounded on x86-64 SSE2 and on gfx950 (hipcc with
 * -ffp-contract=off and the default correctly-rounded fp32 div/sqrt).  They
 * restate the published Cephes single-precision algorithms (Moshier, netlib
 * cephes/single: sinf.c, atanf.c, asinf.c, logf.c, expf.c) with a fixed
 * evaluation order.  No fma, no double, no table lookups.
 *
 * Consumers: the HIP kernels (ezrt_amd/csrc/hip), the host scene-build code
 * (ezrt_amd/csrc/host) and -- as the same *specification* -- the CPU oracle
 * (oracle/).  tests/test_detmath.py checks accuracy against libm and, on the
 * GPU, bit-equality between the host and device evaluations.
 *
 * Compile every consumer with -ffp-contract=off and without fast-math.
 */
#ifndef EZRT_DETMATH_H
#define EZRT_DETMATH_H

#if defined(__HIPCC__)
#define EZ_HD __host__ __device__ __forceinline__
#else
#define EZ_HD static inline
#endif

/* GLSL "#define PI 3.1415926" (P5/fsh:27) rounded to binary32. */
#define EZ_PI 3.1415926f
#define EZ_INF 114514.0f /* P5/fsh:28 */

EZ_HD unsigned ez_f2u(float f) {
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
EZ_HD float ez_u2f(unsigned u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

EZ_HD float ez_abs(float x) { return ez_u2f(ez_f2u(x) & 0x7fffffffu); }
/* GLM/GLSL min/max semantics fixed as (b<a)?b:a / (a<b)?b:a. */
EZ_HD float ez_min(float a, float b) { return (b < a) ? b : a; }
EZ_HD float ez_max(float a, float b) { return (a < b) ? b : a; }
EZ_HD float ez_clamp(float x, float lo, float hi) { return ez_min(ez_max(x, lo), hi); }
EZ_HD float ez_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

/* floor for |x| < 2^31 via truncation (exact). */
EZ_HD float ez_floor(float x) {
  float t = (float)(int)x;
  return (t > x) ? (t - 1.0f) : t;
}

/* x * 2^n for the small n used here, by two exact power-of-two multiplies. */
EZ_HD float ez_ldexp(float x, int n) {
  if (n > 127) {
    x = x * ez_u2f(0x7f000000u); /* 2^127 */
    n -= 127;
    if (n > 127) n = 127;
  } else if (n < -126) {
    x = x * ez_u2f(0x00800000u); /* 2^-126 */
    n += 126;
    if (n < -126) n = -126;
  }
  return x * ez_u2f((unsigned)(n + 127) << 23);
}

/* ---- sin / cos: Cody-Waite pi/2 reduction + Cephes sinf/cosf kernels ------
 * valid for |x| < ~1e4 (the path only feeds [-2pi, 2pi]). */
EZ_HD void ez_sincos(float x, float* s, float* c) {
  const float TWO_OVER_PI = 0.63661977236758134f;
  const float P1 = 1.5703125f;               /* pi/2 split, 3 parts */
  const float P2 = 4.837512969970703125e-4f;
  const float P3 = 7.54978995489188216e-8f;
  float kf = ez_floor(x * TWO_OVER_PI + 0.5f);
  int k = (int)kf;
  float r = ((x - kf * P1) - kf * P2) - kf * P3;
  float z = r * r;
  /* sin(r), |r| <= pi/4 */
  float ps = -1.9515295891e-4f;
  ps = ps * z + 8.3321608736e-3f;
  ps = ps * z - 1.6666654611e-1f;
  float sr = ps * z * r + r;
  /* cos(r) */
  float pc = 2.443315711809948e-5f;
  pc = pc * z - 1.388731625493765e-3f;
  pc = pc * z + 4.166664568298827e-2f;
  float cr = pc * z * z - 0.5f * z + 1.0f;
  int q = k & 3;
  float ss = (q & 1) ? cr : sr;
  float cc = (q & 1) ? sr : cr;
  if (q == 2 || q == 3) ss = -ss;
  if (q == 1 || q == 2) cc = -cc;
  *s = ss;
  *c = cc;
}
EZ_HD float ez_sin(float x) {
  float s, c;
  ez_sincos(x, &s, &c);
  return s;
}
EZ_HD float ez_cos(float x) {
  float s, c;
  ez_sincos(x, &s, &c);
  return c;
}

/* ---- atan (Cephes atanf), x >= 0 kernel ---------------------------------- */
EZ_HD float ez_atan_pos(float x) {
  float y;
  if (x > 2.414213562373095f) { /* tan(3pi/8) */
    y = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
    y = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  float z = x * x;
  float p = 8.05374449538e-2f;
  p = p * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  y = y + (p * z * x + x);
  return y;
}
EZ_HD float ez_atan(float x) { return (x < 0.0f) ? -ez_atan_pos(-x) : ez_atan_pos(x); }

/* GLSL atan(y, x) (P5/fsh:685).  Result in [-pi, pi] (true-pi constants, not
 * EZ_PI: this is the built-in, the shader's PI only appears in its own code). */
EZ_HD float ez_atan2(float y, float x) {
  const float PI_F = 3.14159265358979323846f;
  const float PIO2_F = 1.5707963267948966f;
  if (x == 0.0f) {
    if (y > 0.0f) return PIO2_F;
    if (y < 0.0f) return -PIO2_F;
    return 0.0f;
  }
  float a = ez_atan(y / x);
  if (x < 0.0f) a = (y < 0.0f) ? (a - PI_F) : (a + PI_F);
  return a;
}

/* ---- asin (Cephes asinf); input clamped to [-1, 1] ----------------------- */
EZ_HD float ez_asin(float xx) {
  float a = ez_abs(xx);
  if (a > 1.0f) a = 1.0f;
  float x, z;
  int flag = 0;
  if (a > 0.5f) {
    z = 0.5f * (1.0f - a);
    x = __builtin_sqrtf(z);
    flag = 1;
  } else {
    x = a;
    z = x * x;
  }
  float p = 4.2163199048e-2f;
  p = p * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  float r = p * z * x + x;
  if (flag) {
    r = r + r;
    r = 1.5707963267948966f - r;
  }
  return (xx < 0.0f) ? -r : r;
}

/* ---- log (Cephes logf), x > 0 finite; x <= 0 returns -inf-like large ------ */
EZ_HD float ez_log(float x) {
  if (!(x > 0.0f)) return -ez_u2f(0x7f800000u);
  unsigned u = ez_f2u(x);
  int e = 0;
  if (u < 0x00800000u) { /* subnormal: scale by 2^24 */
    x = x * 16777216.0f;
    u = ez_f2u(x);
    e = -24;
  }
  e += (int)(u >> 23) - 126;                       /* frexp exponent */
  float m = ez_u2f((u & 0x007fffffu) | 0x3f000000u); /* mantissa in [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = p * m - 1.1514610310e-1f;
  p = p * m + 1.1676998740e-1f;
  p = p * m - 1.2420140846e-1f;
  p = p * m + 1.4249322787e-1f;
  p = p * m - 1.6668057665e-1f;
  p = p * m + 2.0000714765e-1f;
  p = p * m - 2.4999993993e-1f;
  p = p * m + 3.3333331174e-1f;
  float y = m * z * p;
  float fe = (float)e;
  y = y + fe * -2.12194440e-4f;
  y = y - 0.5f * z;
  float r = m + y;
  r = r + fe * 0.693359375f;
  return r;
}

/* ---- exp (Cephes expf) ---------------------------------------------------- */
EZ_HD float ez_exp(float x) {
  if (x > 88.72283905206835f) return ez_u2f(0x7f800000u);
  if (x < -103.278929903431851103f) return 0.0f;
  float nf = ez_floor(x * 1.44269504088896341f + 0.5f);
  int n = (int)nf;
  x = x - nf * 0.693359375f;
  x = x - nf * -2.12194440e-4f;
  float z = x * x;
  float p = 1.9875691500e-4f;
  p = p * x + 1.3981999507e-3f;
  p = p * x + 8.3334519073e-3f;
  p = p * x + 4.1665795894e-2f;
  p = p * x + 1.6666665459e-1f;
  p = p * x + 5.0000001201e-1f;
  float r = p * z + x + 1.0f;
  return ez_ldexp(r, n);
}

/* GLSL pow(x, y), x >= 0 (P5/fsh:619; pass3.fsh:22).  Defined as
 * exp(y * log(x)); pow(0, y>0) = 0; pow(x, 0) = 1. */
EZ_HD float ez_pow(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (!(x > 0.0f)) return 0.0f;
  return ez_exp(y * ez_log(x));
}

#endif /* EZRT_DETMATH_H */
