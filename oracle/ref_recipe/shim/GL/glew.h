/* TEST INFRASTRUCTURE (oracle/_ref build only) -- never part of the product.
 *
 * Headless stand-in for <GL/glew.h> + <GL/freeglut.h> so that the reference's main.cpp files
 * compile and RUN here without a GL context.  No rendering happens: every entry point is a
 * no-op, except that the calls which form the reference's data boundary RECORD
 * their payload so a test can read exactly what the reference would have uploaded:
 *   glBufferData(GL_TEXTURE_BUFFER, ...)   -> the encoded triangle / node arrays
 *   glTexImage2D(..., GL_RGB, GL_FLOAT, p) -> hdrMap / hdrCache
 *   glUniform*(glGetUniformLocation(prog, name), ...) -> eye, cameraRotate, frameCounter, ...
 */
#ifndef EZRT_REF_GL_SHIM_H
#define EZRT_REF_GL_SHIM_H
#include <cstddef>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef unsigned int GLuint;
typedef int GLint;
typedef int GLsizei;
typedef unsigned int GLenum;
typedef unsigned int GLbitfield;
typedef unsigned char GLboolean;
typedef float GLfloat;
typedef char GLchar;
typedef void GLvoid;
typedef std::ptrdiff_t GLsizeiptr;
typedef std::ptrdiff_t GLintptr;

enum : unsigned {
    GL_FALSE = 0, GL_TRUE = 1,
    GL_ARRAY_BUFFER = 0x8892, GL_ELEMENT_ARRAY_BUFFER = 0x8893, GL_TEXTURE_BUFFER = 0x8C2A,
    GL_STATIC_DRAW = 0x88E4, GL_COMPILE_STATUS = 0x8B81, GL_VERTEX_SHADER = 0x8B31, GL_FRAGMENT_SHADER = 0x8B30,
    GL_UNSIGNED_INT = 0x1405, GL_FLOAT = 0x1406, GL_TRIANGLES = 4, GL_LINES = 1, GL_LINE = 0x1B01,
    GL_FRONT_AND_BACK = 0x0408, GL_DEPTH_TEST = 0x0B71, GL_DEPTH_BUFFER_BIT = 0x100, GL_COLOR_BUFFER_BIT = 0x4000,
    GL_TEXTURE_2D = 0x0DE1, GL_FRAMEBUFFER = 0x8D40, GL_RGB32F = 0x8815, GL_RGBA32F = 0x8814, GL_RGB = 0x1907,
    GL_RGBA = 0x1908, GL_NEAREST = 0x2600, GL_LINEAR = 0x2601, GL_CLAMP_TO_EDGE = 0x812F,
    GL_TEXTURE_MIN_FILTER = 0x2801, GL_TEXTURE_MAG_FILTER = 0x2800, GL_TEXTURE_WRAP_S = 0x2802, GL_TEXTURE_WRAP_T = 0x2803,
    GL_TEXTURE0 = 0x84C0, GL_TEXTURE1, GL_TEXTURE2, GL_TEXTURE3, GL_TEXTURE4, GL_TEXTURE5,
    GL_COLOR_ATTACHMENT0 = 0x8CE0,
    GLUT_RGBA = 0, GLUT_DEPTH = 16, GLUT_LEFT_BUTTON = 0, GLUT_DOWN = 0
};

struct EzrtRefGLRecord {
    struct Blob { std::vector<float> data; int w = 0, h = 0; };
    std::vector<Blob> texture_buffers;      /* glBufferData(GL_TEXTURE_BUFFER) in call order */
    std::vector<Blob> rgb_images;           /* glTexImage2D(GL_RGB, GL_FLOAT, non-null) in call order */
    std::map<std::string, std::vector<float>> uniforms_f;
    std::map<std::string, long long> uniforms_i;
    std::vector<std::string> names{""};    /* uniform location -> name */
    std::vector<int> filters;               /* GL_TEXTURE_MIN_FILTER values in call order */
    GLuint next_id = 1;
    bool record_images = true;   /* chapter 3's main() passes an uninitialised HDRLoaderResult */
    void clear() { *this = EzrtRefGLRecord(); }
};
inline EzrtRefGLRecord g_ezrt_ref_gl;

inline void glGenBuffers(GLsizei n, GLuint* o) { for (int i = 0; i < n; i++) o[i] = g_ezrt_ref_gl.next_id++; }
inline void glGenTextures(GLsizei n, GLuint* o) { glGenBuffers(n, o); }
inline void glGenVertexArrays(GLsizei n, GLuint* o) { glGenBuffers(n, o); }
inline void glGenFramebuffers(GLsizei n, GLuint* o) { glGenBuffers(n, o); }
inline void glBufferData(GLenum target, GLsizeiptr size, const void* p, GLenum) {
    if (target == GL_TEXTURE_BUFFER && p) {
        EzrtRefGLRecord::Blob b;
        b.data.resize(size_t(size) / sizeof(float));
        std::memcpy(b.data.data(), p, b.data.size() * sizeof(float));
        g_ezrt_ref_gl.texture_buffers.push_back(b);
    }
}
inline void glTexImage2D(GLenum, GLint, GLint, GLsizei w, GLsizei h, GLint, GLenum format, GLenum type, const void* p) {
    if (g_ezrt_ref_gl.record_images && format == GL_RGB && type == GL_FLOAT && p && w > 0 && h > 0 && w <= 16384 && h <= 16384) {
        EzrtRefGLRecord::Blob b;
        b.w = w; b.h = h;
        b.data.resize(size_t(w) * h * 3);
        std::memcpy(b.data.data(), p, b.data.size() * sizeof(float));
        g_ezrt_ref_gl.rgb_images.push_back(b);
    }
}
inline void glTexParameteri(GLenum, GLenum pname, GLint v) { if (pname == GL_TEXTURE_MIN_FILTER) g_ezrt_ref_gl.filters.push_back(v); }
inline GLint glGetUniformLocation(GLuint, const GLchar* name) {
    g_ezrt_ref_gl.names.push_back(name);
    return GLint(g_ezrt_ref_gl.names.size() - 1);
}
inline void glUniform1i(GLint loc, GLint v) { g_ezrt_ref_gl.uniforms_i[g_ezrt_ref_gl.names[loc]] = v; }
inline void glUniform1ui(GLint loc, GLuint v) { g_ezrt_ref_gl.uniforms_i[g_ezrt_ref_gl.names[loc]] = v; }
inline void glUniform3fv(GLint loc, GLsizei, const GLfloat* v) { g_ezrt_ref_gl.uniforms_f[g_ezrt_ref_gl.names[loc]].assign(v, v + 3); }
inline void glUniformMatrix4fv(GLint loc, GLsizei, GLboolean, const GLfloat* v) { g_ezrt_ref_gl.uniforms_f[g_ezrt_ref_gl.names[loc]].assign(v, v + 16); }
inline void glGetShaderiv(GLuint, GLenum, GLint* out) { *out = 1; }
inline GLuint glCreateShader(GLenum) { return g_ezrt_ref_gl.next_id++; }
inline GLuint glCreateProgram() { return g_ezrt_ref_gl.next_id++; }
inline int glewInit() { return 0; }

#define EZRT_GL_NOOP(name) template <class... A> inline void name(A...) {}
EZRT_GL_NOOP(glBindBuffer) EZRT_GL_NOOP(glBufferSubData) EZRT_GL_NOOP(glBindTexture) EZRT_GL_NOOP(glTexBuffer)
EZRT_GL_NOOP(glActiveTexture) EZRT_GL_NOOP(glUseProgram) EZRT_GL_NOOP(glBindFramebuffer) EZRT_GL_NOOP(glBindVertexArray)
EZRT_GL_NOOP(glShaderSource) EZRT_GL_NOOP(glCompileShader) EZRT_GL_NOOP(glGetShaderInfoLog) EZRT_GL_NOOP(glDeleteShader)
EZRT_GL_NOOP(glAttachShader) EZRT_GL_NOOP(glLinkProgram) EZRT_GL_NOOP(glViewport) EZRT_GL_NOOP(glVertexAttribPointer)
EZRT_GL_NOOP(glEnableVertexAttribArray) EZRT_GL_NOOP(glEnable) EZRT_GL_NOOP(glDrawBuffers) EZRT_GL_NOOP(glDrawArrays)
EZRT_GL_NOOP(glDrawElements) EZRT_GL_NOOP(glClearColor) EZRT_GL_NOOP(glClear) EZRT_GL_NOOP(glFramebufferTexture2D)
EZRT_GL_NOOP(glPolygonMode)
EZRT_GL_NOOP(glutInit) EZRT_GL_NOOP(glutInitDisplayMode) EZRT_GL_NOOP(glutInitWindowSize) EZRT_GL_NOOP(glutInitWindowPosition)
EZRT_GL_NOOP(glutCreateWindow) EZRT_GL_NOOP(glutDisplayFunc) EZRT_GL_NOOP(glutIdleFunc) EZRT_GL_NOOP(glutMotionFunc)
EZRT_GL_NOOP(glutMouseFunc) EZRT_GL_NOOP(glutMouseWheelFunc) EZRT_GL_NOOP(glutMainLoop) EZRT_GL_NOOP(glutSwapBuffers)
EZRT_GL_NOOP(glutPostRedisplay)
#undef EZRT_GL_NOOP
#endif
