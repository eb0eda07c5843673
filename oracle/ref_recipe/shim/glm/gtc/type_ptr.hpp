/* TEST INFRASTRUCTURE (oracle/_ref build only).  `<glm/gtc/type_ptr.hpp>` stand-in. */
#ifndef EZRT_REF_GLM_SHIM_TP_HPP
#define EZRT_REF_GLM_SHIM_TP_HPP
#include "../glm.hpp"
namespace glm {
inline const float* value_ptr(const vec3& v) { return &v.x; }
inline const float* value_ptr(const vec4& v) { return &v.x; }
inline const float* value_ptr(const mat4& m) { return &m.c[0].x; }
}  // namespace glm
#endif
