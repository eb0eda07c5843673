/* TEST INFRASTRUCTURE (oracle/_ref build only).  `<glm/gtc/matrix_transform.hpp>` stand-in:
 * translate / rotate / scale / lookAt (right-handed, GLM's default) restated from the published
 * GLM 0.9.9.x ext/matrix_transform.inl -- see glm.hpp in this directory. */
#ifndef EZRT_REF_GLM_SHIM_MT_HPP
#define EZRT_REF_GLM_SHIM_MT_HPP
#include "../glm.hpp"
namespace glm {
inline mat4 translate(const mat4& m, const vec3& v) {
    mat4 R(m);
    R[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3];
    return R;
}
inline mat4 scale(const mat4& m, const vec3& v) {
    mat4 R;
    R[0] = m[0] * v[0]; R[1] = m[1] * v[1]; R[2] = m[2] * v[2]; R[3] = m[3];
    return R;
}
inline mat4 rotate(const mat4& m, float angle, const vec3& v) {
    const float a = angle;
    const float c = std::cos(a);
    const float s = std::sin(a);
    vec3 axis(normalize(v));
    vec3 temp((1.0f - c) * axis);
    float R00 = c + temp[0] * axis[0];
    float R01 = temp[0] * axis[1] + s * axis[2];
    float R02 = temp[0] * axis[2] - s * axis[1];
    float R10 = temp[1] * axis[0] - s * axis[2];
    float R11 = c + temp[1] * axis[1];
    float R12 = temp[1] * axis[2] + s * axis[0];
    float R20 = temp[2] * axis[0] + s * axis[1];
    float R21 = temp[2] * axis[1] - s * axis[0];
    float R22 = c + temp[2] * axis[2];
    mat4 Result;
    Result[0] = m[0] * R00 + m[1] * R01 + m[2] * R02;
    Result[1] = m[0] * R10 + m[1] * R11 + m[2] * R12;
    Result[2] = m[0] * R20 + m[1] * R21 + m[2] * R22;
    Result[3] = m[3];
    return Result;
}
inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up) {
    const vec3 f(normalize(center - eye));
    const vec3 s(normalize(cross(f, up)));
    const vec3 u(cross(s, f));
    mat4 R(1.0f);
    R[0][0] = s.x; R[1][0] = s.y; R[2][0] = s.z;
    R[0][1] = u.x; R[1][1] = u.y; R[2][1] = u.z;
    R[0][2] = -f.x; R[1][2] = -f.y; R[2][2] = -f.z;
    R[3][0] = -dot(s, eye); R[3][1] = -dot(u, eye); R[3][2] = dot(f, eye);
    return R;
}
}  // namespace glm
#endif
