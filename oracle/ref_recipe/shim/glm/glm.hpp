/* TEST INFRASTRUCTURE (oracle/_ref build only) -- never part of the product.
 *
 * Minimal stand-in for the GLM headers the reference's main.cpp files include
 * (`<glm/glm.hpp>`, P2/P3/P4/P5 main.cpp:11-15).  GLM is a third-party dependency that is
 * NOT vendored under /root/reference and not installed in this image, so its arithmetic is
 * restated here from the published GLM 0.9.9.x sources (the reference's CMakeLists pins no
 * version): only the types and functions those main.cpp files use, with GLM's evaluation order
 *   dot(a,b)      = (a*b).x + (a*b).y + (a*b).z              (detail/func_geometric.inl)
 *   normalize(v)  = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
 *   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
 *   min(x,y)      = (y < x) ? y : x ;  max(x,y) = (x < y) ? y : x   (detail/func_common.inl)
 *   radians(d)    = d * 0.01745329251994329576923690768489
 *   mat4 * vec4   = (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)  (detail/type_mat4x4.inl)
 *   mat4 * mat4   : column j = A[0]*B[j][0] + A[1]*B[j][1] + A[2]*B[j][2] + A[3]*B[j][3]
 * Everything the reference computes AROUND these calls (readObj, the builders, hitTriangle,
 * calculateHdrCache, the encode loops, display()'s camera) is the reference's own code,
 * compiled from where it lies.  Build with -ffp-contract=off (GLM results on the author's
 * x86/MSVC build had no fma contraction either).
 */
#ifndef EZRT_REF_GLM_SHIM_HPP
#define EZRT_REF_GLM_SHIM_HPP
#include <cmath>
#include <cstddef>

namespace glm {

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    template <class A> explicit vec3(A s) : x(float(s)), y(float(s)), z(float(s)) {}
    template <class A, class B, class C> vec3(A a, B b, C c) : x(float(a)), y(float(b)), z(float(c)) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    vec3& operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3& a, const vec3& b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }

struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    template <class A, class B, class C, class D> vec4(A a, B b, C c, D d) : x(float(a)), y(float(b)), z(float(c)), w(float(d)) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec4 operator+(const vec4& a, const vec4& b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4& a, const vec4& b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4& a, const vec4& b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(const vec4& a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline vec4 operator*(float s, const vec4& a) { return vec4(s * a.x, s * a.y, s * a.z, s * a.w); }

struct mat4 {          /* column-major, c[j] = column j (GLM: m[j]) */
    vec4 c[4];
    mat4() { c[0] = vec4(1, 0, 0, 0); c[1] = vec4(0, 1, 0, 0); c[2] = vec4(0, 0, 1, 0); c[3] = vec4(0, 0, 0, 1); }
    explicit mat4(float s) { c[0] = vec4(s, 0, 0, 0); c[1] = vec4(0, s, 0, 0); c[2] = vec4(0, 0, s, 0); c[3] = vec4(0, 0, 0, s); }
    mat4(const vec4& a, const vec4& b, const vec4& d, const vec4& e) { c[0] = a; c[1] = b; c[2] = d; c[3] = e; }
    vec4& operator[](int j) { return c[j]; }
    const vec4& operator[](int j) const { return c[j]; }
};
inline vec4 operator*(const mat4& m, const vec4& v) {
    const vec4 Mul0 = m[0] * vec4(v.x, v.x, v.x, v.x);
    const vec4 Mul1 = m[1] * vec4(v.y, v.y, v.y, v.y);
    const vec4 Add0 = Mul0 + Mul1;
    const vec4 Mul2 = m[2] * vec4(v.z, v.z, v.z, v.z);
    const vec4 Mul3 = m[3] * vec4(v.w, v.w, v.w, v.w);
    const vec4 Add1 = Mul2 + Mul3;
    return Add0 + Add1;
}
inline mat4 operator*(const mat4& A, const mat4& B) {
    mat4 R;
    for (int j = 0; j < 4; j++)
        R[j] = A[0] * B[j][0] + A[1] * B[j][1] + A[2] * B[j][2] + A[3] * B[j][3];
    return R;
}

template <class T> inline T min(T x, T y) { return (y < x) ? y : x; }
template <class T> inline T max(T x, T y) { return (x < y) ? y : x; }
inline vec3 min(const vec3& x, const vec3& y) { return vec3(min(x.x, y.x), min(x.y, y.y), min(x.z, y.z)); }
inline vec3 max(const vec3& x, const vec3& y) { return vec3(max(x.x, y.x), max(x.y, y.y), max(x.z, y.z)); }
template <class T> inline T sin(T v) { return std::sin(v); }
template <class T> inline T cos(T v) { return std::cos(v); }
template <class T> inline T radians(T d) { return d * static_cast<T>(0.01745329251994329576923690768489); }

inline float dot(const vec3& a, const vec3& b) { vec3 t(a * b); return t.x + t.y + t.z; }
inline float dot(const vec4& a, const vec4& b) { vec4 t(a * b); return (t.x + t.y) + (t.z + t.w); }
inline vec3 cross(const vec3& x, const vec3& y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
inline float length(const vec3& v) { return std::sqrt(dot(v, v)); }
inline vec3 normalize(const vec3& v) { return v * inversesqrt(dot(v, v)); }

/* detail/func_matrix.inl compute_inverse<4,4>: cofactor expansion, GLM's operand order */
inline mat4 inverse(const mat4& m) {
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 Fac0(Coef00, Coef00, Coef02, Coef03);
    vec4 Fac1(Coef04, Coef04, Coef06, Coef07);
    vec4 Fac2(Coef08, Coef08, Coef10, Coef11);
    vec4 Fac3(Coef12, Coef12, Coef14, Coef15);
    vec4 Fac4(Coef16, Coef16, Coef18, Coef19);
    vec4 Fac5(Coef20, Coef20, Coef22, Coef23);
    vec4 Vec0(m[1][0], m[0][0], m[0][0], m[0][0]);
    vec4 Vec1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 Vec2(m[1][2], m[0][2], m[0][2], m[0][2]);
    vec4 Vec3(m[1][3], m[0][3], m[0][3], m[0][3]);
    vec4 Inv0(Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2);
    vec4 Inv1(Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4);
    vec4 Inv2(Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5);
    vec4 Inv3(Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5);
    vec4 SignA(+1, -1, +1, -1);
    vec4 SignB(-1, +1, -1, +1);
    mat4 Inverse(Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB);
    vec4 Row0(Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]);
    vec4 Dot0(m[0] * Row0);
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    return mat4(Inverse[0] * OneOverDeterminant, Inverse[1] * OneOverDeterminant,
                Inverse[2] * OneOverDeterminant, Inverse[3] * OneOverDeterminant);
}

}  // namespace glm
#endif
