/* TEST INFRASTRUCTURE -- oracle/ref_recipe/shim/glsl_shim.h
 *
 * A stand-in for the GLSL 3.30/4.30 language subset the reference's three fragment shaders use
 * (P3/P4/P5 shaders/fshader.fsh), so that the shader text ITSELF can be compiled by g++ and executed
 * per pixel-sample as the pin of the shader half of the oracle.  Nothing of the
 * reference is restated here: this header only supplies the language -- vector types, swizzles,
 * constructors, operators, samplers and built-in functions.  wrap_fsh.cpp #includes the shader (after
 * the mechanical source pass of fsh_pass.py) inside `struct Fsh : GlslBuiltins { ... }`.
 *
 * Three things GLSL leaves to the driver are DEFINED here, exactly as in the rest of this repository
 * (SURVEY.md 2.3 / 8c "restatement rules that are definitions"):
 *   - the precision of sin cos atan asin log pow: include/ezrt_detmath.h;
 *   - the expansion of the vector built-ins: dot = ax*bx + ay*by + az*bz, normalize(v) = v * (1/sqrt(dot(v,v))),
 *     min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a, mix(x,y,a) = x*(1-a) + y*a, reflect(I,N) = I - 2*dot(N,I)*N,
 *     mat4 * vec4 = ((c0*x + c1*y) + c2*z) + c3*w, clamp = min(max(x,lo),hi);
 *   - texture filtering: texel centres, clamp-to-edge, NEAREST = floor(u*W), BILINEAR = the GL formula in fp32
 *     (x-lerp, then y-lerp), NaN coordinates read texel 0.
 * Everything else -- which operations the shader performs, in which order, on which operands -- comes from the
 * reference's text.  Compile with -ffp-contract=off -fno-fast-math -fsingle-precision-constant (a GLSL literal
 * `1.0` is a float).
 */
#ifndef EZRT_GLSL_SHIM_H
#define EZRT_GLSL_SHIM_H

#include <stddef.h>
#include <stdint.h>

#include "ezrt_detmath.h"

typedef unsigned int uint;

struct vec2 {
  union { float x, r; };
  union { float y, g; };
  vec2() {}
  explicit vec2(float s) : x(s), y(s) {}
  vec2(float a, float b) : x(a), y(b) {}
  vec2& operator+=(float s) { x = x + s; y = y + s; return *this; }
  vec2& operator/=(const vec2& o) { x = x / o.x; y = y / o.y; return *this; }
};
inline vec2 operator+(const vec2& a, const vec2& b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator+(const vec2& a, float s) { return vec2(a.x + s, a.y + s); }
inline vec2 operator*(const vec2& a, float s) { return vec2(a.x * s, a.y * s); }

struct vec3 {
  union { float x, r; };
  union { float y, g; };
  union { float z, b; };
  vec3() {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  vec3(float a, float b_, float c) : x(a), y(b_), z(c) {}
  vec2 xy() const { return vec2(x, y); }
  vec3& operator+=(const vec3& o) { x = x + o.x; y = y + o.y; z = z + o.z; return *this; }
  vec3& operator*=(const vec3& o) { x = x * o.x; y = y * o.y; z = z * o.z; return *this; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3& a, const vec3& b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator/(float s, const vec3& a) { return vec3(s / a.x, s / a.y, s / a.z); }

struct vec4 {
  union { float x, r; };
  union { float y, g; };
  union { float z, b; };
  union { float w, a; };
  vec4() {}
  vec4(float a_, float b_, float c, float d) : x(a_), y(b_), z(c), w(d) {}
  vec4(const vec2& v, float c, float d) : x(v.x), y(v.y), z(c), w(d) {}
  vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
  vec3 xyz() const { return vec3(x, y, z); }
  vec3 rgb() const { return vec3(x, y, z); }
  vec2 rg() const { return vec2(x, y); }
  vec2 xy() const { return vec2(x, y); }
};

struct ivec3 {
  int x, y, z;
  explicit ivec3(const vec3& v) : x((int)v.x), y((int)v.y), z((int)v.z) {} /* GLSL float -> int: truncation */
};

struct mat4 { /* column-major, as the uniform is uploaded (GL_FALSE) */
  float m[16];
};
inline vec4 operator*(const mat4& M, const vec4& v) {
  const float* m = M.m;
  return vec4(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w, ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w,
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w, ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w);
}

/* samplerBuffer over an RGB32F buffer texture (GL_RGB32F, P5/main.cpp:881-893): texel i = floats 3i .. 3i+2 */
struct samplerBuffer {
  const float* data = nullptr;
};
/* sampler2D over an RGB32F image, row 0 = v 0 */
struct sampler2D {
  const float* data = nullptr;
  int W = 0, H = 0;
  int bilinear = 0;
};

struct GlslBuiltins {
  static float abs(float x) { return ez_abs(x); }
  static float sqrt(float x) { return __builtin_sqrtf(x); }
  static float sin(float x) { return ez_sin(x); }
  static float cos(float x) { return ez_cos(x); }
  static float atan(float y, float x) { return ez_atan2(y, x); }
  static float asin(float x) { return ez_asin(x); }
  static float log(float x) { return ez_log(x); }
  static float pow(float x, float y) { return ez_pow(x, y); }
  static float min(float a, float b) { return ez_min(a, b); }
  static float max(float a, float b) { return ez_max(a, b); }
  static vec3 min(const vec3& a, const vec3& b) { return vec3(ez_min(a.x, b.x), ez_min(a.y, b.y), ez_min(a.z, b.z)); }
  static vec3 max(const vec3& a, const vec3& b) { return vec3(ez_max(a.x, b.x), ez_max(a.y, b.y), ez_max(a.z, b.z)); }
  static float clamp(float x, float lo, float hi) { return ez_clamp(x, lo, hi); }
  static float mix(float x, float y, float a) { return ez_mix(x, y, a); }
  static vec3 mix(const vec3& x, const vec3& y, float a) { return vec3(ez_mix(x.x, y.x, a), ez_mix(x.y, y.y, a), ez_mix(x.z, y.z, a)); }
  static float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
  static vec3 cross(const vec3& a, const vec3& b) {
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
  }
  static vec3 normalize(const vec3& a) {
    const float inv = 1.0f / __builtin_sqrtf(dot(a, a));
    return a * inv;
  }
  static vec3 reflect(const vec3& I, const vec3& N) {
    const float k = 2.0f * dot(N, I);
    return I - N * k;
  }
  static vec4 texelFetch(const samplerBuffer& s, int i) {
    const float* p = s.data + (size_t)i * 3;
    return vec4(p[0], p[1], p[2], 1.0f);
  }
  static float sane01(float u) {
    if (!(u == u)) return 0.0f;
    return ez_clamp(u, 0.0f, 1.0f);
  }
  static vec3 texel(const sampler2D& s, int ix, int iy) {
    const float* p = s.data + ((size_t)iy * s.W + ix) * 3;
    return vec3(p[0], p[1], p[2]);
  }
  static vec4 texture2D(const sampler2D& s, const vec2& uv) {
    const float u = sane01(uv.x), v = sane01(uv.y);
    const int W = s.W, H = s.H;
    if (!s.bilinear) {
      int ix = (int)ez_floor(u * (float)W), iy = (int)ez_floor(v * (float)H);
      if (ix > W - 1) ix = W - 1;
      if (iy > H - 1) iy = H - 1;
      return vec4(texel(s, ix, iy), 1.0f);
    }
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0 = ez_floor(x), y0 = ez_floor(y);
    const float fx = x - x0, fy = y - y0;
    int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    if (ix0 < 0) ix0 = 0;
    if (iy0 < 0) iy0 = 0;
    if (ix1 > W - 1) ix1 = W - 1;
    if (iy1 > H - 1) iy1 = H - 1;
    const vec3 top = mix(texel(s, ix0, iy0), texel(s, ix1, iy0), fx);
    const vec3 bot = mix(texel(s, ix0, iy1), texel(s, ix1, iy1), fx);
    return vec4(mix(top, bot, fy), 1.0f);
  }
};

#endif
