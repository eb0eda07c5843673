// ezrt_scene.hpp -- host-side scene-build API (C++), the mirror of what EzRT's
// main.cpp files do before the first frame: same structs, same function names,
// same argument meaning, results-compatible semantics (including the quirks
// listed in SURVEY.md 8a).  New code; GLM is replaced by the small PODs below
// whose evaluation order is fixed here (GLM is un-vendored in the reference, so
// its arithmetic is "parity unpinned" -- these definitions are the spec).
//
//   reference interface                                         file:line
//   struct Material / Triangle / BVHNode                        P3/main.cpp:28-57 (P4 defaults P4/main.cpp:27-43)
//   struct Triangle_encoded / BVHNode_encoded                   P3/main.cpp:61-76
//   mat4 getTransformMatrix(rot, trans, scale)                  P3/main.cpp:254-270
//   void readObj(path, triangles, material, trans, smooth)      P3/main.cpp:273-391
//   int  buildBVH(triangles, nodes, l, r, n)                    P3/main.cpp:394-454
//   int  buildBVHwithSAH(triangles, nodes, l, r, n)             P3/main.cpp:457-588
//   encode loops ("encodeTriangle" / "encodeBVH")               P3/main.cpp:720-748
//   HDRLoader::load(fileName, HDRLoaderResult&)                 P5/lib/hdrloader.h:10-20, hdrloader.cpp:50-118
//   float* calculateHdrCache(HDR, width, height)                P5/main.cpp:592-689
//   eye / cameraRotate from (rotatAngle, upAngle, r)            P3/main.cpp:607-610
//
// Error behaviour: the reference prints and exit(-1)s on unreadable files
// (P3/main.cpp:282-285) and ignores HDRLoader's bool; here readObj throws
// std::runtime_error and HDRLoader::load returns false -- a library must not
// exit.
#ifndef EZRT_SCENE_HPP
#define EZRT_SCENE_HPP

#include <cstdint>
#include <string>
#include <vector>

namespace ezrt {

struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct vec4 {
  float x, y, z, w;
};
// column-major like glm::mat4: c[col][row]
struct mat4 {
  float c[4][4];
};

mat4 identity();
mat4 translate(const mat4& m, vec3 v);
mat4 scale(const mat4& m, vec3 v);
mat4 rotate(const mat4& m, float angle_rad, vec3 axis);
mat4 lookAt(vec3 eye, vec3 center, vec3 up); // right-handed
mat4 inverse(const mat4& m);
mat4 mul(const mat4& a, const mat4& b);
vec4 mul(const mat4& m, vec4 v);
float radians(float deg);

// P3/main.cpp:28-43.  Defaults are P3's; chapter 4/5 defaults via disneyDefaults().
struct Material {
  vec3 emissive = vec3(0, 0, 0);
  vec3 baseColor = vec3(1, 1, 1);
  float subsurface = 0.0f;
  float metallic = 0.0f;
  float specular = 0.0f;
  float specularTint = 0.0f;
  float roughness = 0.0f;
  float anisotropic = 0.0f;
  float sheen = 0.0f;
  float sheenTint = 0.0f;
  float clearcoat = 0.0f;
  float clearcoatGloss = 0.0f;
  float IOR = 1.0f;
  float transmission = 0.0f;
};
Material disneyDefaults(); // P4/main.cpp:27-43: specular .5, roughness .5, sheenTint .5, clearcoatGloss 1

struct Triangle { // P3/main.cpp:46-50
  vec3 p1, p2, p3;
  vec3 n1, n2, n3;
  Material material;
};

struct BVHNode { // P3/main.cpp:53-57
  int left, right;
  int n, index;
  vec3 AA, BB;
};

struct Triangle_encoded { // P3/main.cpp:61-72 (12 x vec3 = 144 B)
  vec3 p1, p2, p3;
  vec3 n1, n2, n3;
  vec3 emissive;
  vec3 baseColor;
  vec3 param1; // (subsurface, metallic, specular)
  vec3 param2; // (specularTint, roughness, anisotropic)
  vec3 param3; // (sheen, sheenTint, clearcoat)
  vec3 param4; // (clearcoatGloss, IOR, transmission)
};
struct BVHNode_encoded { // P3/main.cpp:74-78 (4 x vec3 = 48 B)
  vec3 childs;   // (left, right, 0)
  vec3 leafInfo; // (n, index, 0)
  vec3 AA, BB;
};
static_assert(sizeof(Triangle_encoded) == 144, "reference record size");
static_assert(sizeof(BVHNode_encoded) == 48, "reference record size");

mat4 getTransformMatrix(vec3 rotateCtrl, vec3 translateCtrl, vec3 scaleCtrl);

void readObj(const std::string& filepath, std::vector<Triangle>& triangles, Material material, mat4 trans,
             bool smoothNormal);
// same, from an in-memory OBJ text (what readObj does after opening the file)
void readObjText(const char* text, size_t len, std::vector<Triangle>& triangles, Material material, mat4 trans,
                 bool smoothNormal);

// The node-0 sentinel every main() seeds `nodes` with (P3/main.cpp:707-713).
BVHNode testNode();

int buildBVH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n);
int buildBVHwithSAH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n);

// Order of triangles whose sort keys (centroid coordinates) are exactly equal.  The reference calls
// std::sort, which leaves it to the C++ library (parity unpinned):
//   Stable     (default) equal keys keep their current order -- platform independent, and what the GPU
//              builder (ezrt_build_sah, LSD radix sort) produces;
//   LibrarySort  this toolchain's std::sort on the same keys and comparator: its control flow depends
//              on comparison results only, so the permutation is the one the reference's std::sort of
//              144-byte structs produces when built with the same library (checked against
//              oracle/_ref = the reference compiled here, tests/test_ref_pin.py).
// Per thread, like lastBuildStats().
enum class TieOrder { Stable = 0, LibrarySort = 1 };
void setTieOrder(TieOrder t);
TieOrder tieOrder();

// statistics of the last buildBVHwithSAH call on this thread
struct BuildStats {
  int64_t inf_cap_nodes = 0; // inner nodes whose every SAH candidate cost >= INF (median-x fallback)
  int64_t sorts = 0;
  int max_depth = 0;
};
BuildStats lastBuildStats();

Triangle_encoded encodeTriangle(const Triangle& t);
BVHNode_encoded encodeBVH(const BVHNode& n);
std::vector<Triangle_encoded> encodeTriangles(const std::vector<Triangle>& triangles);
std::vector<BVHNode_encoded> encodeBVH(const std::vector<BVHNode>& nodes);

struct HDRLoaderResult { // P5/lib/hdrloader.h:10-15
  int width = 0, height = 0;
  float* cols = nullptr; // width*height*3, new[]-allocated like the reference; caller delete[]s
};
struct HDRLoader {
  static bool load(const char* fileName, HDRLoaderResult& res);
  static bool loadMemory(const unsigned char* data, size_t len, HDRLoaderResult& res);
};

// returns new float[width*height*3] (caller delete[]s; the reference leaks it)
float* calculateHdrCache(const float* HDR, int width, int height);

struct Camera {
  vec3 eye;
  mat4 cameraRotate;
};
Camera cameraFromAngles(float rotatAngleDeg, float upAngleDeg, float r);

// ---------------------------------------------------------------------------
// Chapter 2's CPU query API on a pointer tree (P2/main.cpp), kept for callers
// that still use it.  Same names and argument meaning; the arithmetic is the
// fp32 contract of DESIGN.md section 2, so distances are bit-identical to the
// GPU trace's.
//
//   struct BVHNode{left,right,n,index,AA,BB}                   P2/main.cpp:28-34
//   struct Triangle{p1,p2,p3,center}                           P2/main.cpp:36-43
//   struct HitResult{Triangle* triangle = NULL; distance=INF}  P2/main.cpp:58-61
//   BVHNode* buildBVH / buildBVHwithSAH(triangles, l, r, n)    P2/main.cpp:242-294, 297-423
//   float hitTriangle(Triangle*, Ray)  (t or INF)              P2/main.cpp:212-238
//   float hitAABB(Ray, AA, BB)                                 P2/main.cpp:449-463
//   HitResult hitTriangleArray(ray, triangles, l, r)           P2/main.cpp:436-446
//   HitResult hitBVH(ray, triangles, root)                     P2/main.cpp:466-485
//
// Deliberate difference: a leaf scans [index, index+n-1] as chapters 3-5 do
// (P3/fsh:334-337); P2/main.cpp:471 passes (n, n+index-1), which only works
// on its demo by accident.  The tree is owned by the caller:
// release it with freeBVH (the reference leaks it).
namespace p2 {
constexpr float INF = 114514.0f;
struct Triangle {
  vec3 p1, p2, p3;
  vec3 center;
  Triangle(vec3 a, vec3 b, vec3 c);
};
struct BVHNode {
  BVHNode* left = nullptr;
  BVHNode* right = nullptr;
  int n = 0, index = 0;
  vec3 AA, BB;
};
struct Ray {
  vec3 startPoint, direction;
};
struct HitResult {
  Triangle* triangle = nullptr;
  float distance = INF;
};
BVHNode* buildBVH(std::vector<Triangle>& triangles, int l, int r, int n);
BVHNode* buildBVHwithSAH(std::vector<Triangle>& triangles, int l, int r, int n);
void freeBVH(BVHNode* root);
float hitTriangle(Triangle* triangle, Ray ray);
float hitAABB(Ray r, vec3 AA, vec3 BB);
HitResult hitTriangleArray(Ray ray, std::vector<Triangle>& triangles, int l, int r);
HitResult hitBVH(Ray ray, std::vector<Triangle>& triangles, BVHNode* root);
} // namespace p2

} // namespace ezrt

#endif
