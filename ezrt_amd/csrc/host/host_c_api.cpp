// host_c_api.cpp -- C shim over ezrt_scene.hpp (see include/ezrt_scene_c.h).
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "ezrt_scene.hpp"
#include "ezrt_scene_c.h"

using namespace ezrt;

struct EzrtHostScene {
  std::vector<Triangle> triangles;
  std::vector<BVHNode> nodes;
  BuildStats stats;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}
#define EZ_TRY try {
#define EZ_CATCH                                   \
  }                                                \
  catch (const std::exception& e) {                \
    return fail(-1, e.what());                     \
  }                                                \
  catch (...) {                                    \
    return fail(-1, "unknown C++ exception");      \
  }

static Material mat_from18(const float* m) {
  Material r;
  r.emissive = vec3(m[0], m[1], m[2]);
  r.baseColor = vec3(m[3], m[4], m[5]);
  r.subsurface = m[6];
  r.metallic = m[7];
  r.specular = m[8];
  r.specularTint = m[9];
  r.roughness = m[10];
  r.anisotropic = m[11];
  r.sheen = m[12];
  r.sheenTint = m[13];
  r.clearcoat = m[14];
  r.clearcoatGloss = m[15];
  r.IOR = m[16];
  r.transmission = m[17];
  return r;
}
static void mat_to18(const Material& r, float* m) {
  m[0] = r.emissive.x; m[1] = r.emissive.y; m[2] = r.emissive.z;
  m[3] = r.baseColor.x; m[4] = r.baseColor.y; m[5] = r.baseColor.z;
  m[6] = r.subsurface; m[7] = r.metallic; m[8] = r.specular; m[9] = r.specularTint;
  m[10] = r.roughness; m[11] = r.anisotropic; m[12] = r.sheen; m[13] = r.sheenTint;
  m[14] = r.clearcoat; m[15] = r.clearcoatGloss; m[16] = r.IOR; m[17] = r.transmission;
}
static mat4 mat4_from16(const float* p) {
  mat4 m;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) m.c[c][r] = p[c * 4 + r];
  return m;
}
static void mat4_to16(const mat4& m, float* p) {
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) p[c * 4 + r] = m.c[c][r];
}

extern "C" {

EzrtHostScene* ezrt_host_scene_new(void) {
  try {
    return new EzrtHostScene();
  } catch (...) {
    return nullptr;
  }
}
void ezrt_host_scene_free(EzrtHostScene* h) { delete h; }

int ezrt_host_material_defaults(int which, float out18[18]) {
  if (!out18) return fail(-1, "NULL argument");
  Material m = (which == 3) ? Material() : disneyDefaults();
  mat_to18(m, out18);
  return 0;
}

int ezrt_host_get_transform_matrix(const float rot[3], const float tr[3], const float sc[3], float out16[16]) {
  if (!rot || !tr || !sc || !out16) return fail(-1, "NULL argument");
  mat4_to16(getTransformMatrix(vec3(rot[0], rot[1], rot[2]), vec3(tr[0], tr[1], tr[2]), vec3(sc[0], sc[1], sc[2])),
            out16);
  return 0;
}

int ezrt_host_read_obj(EzrtHostScene* h, const char* path, const float material18[18], const float trans16[16],
                       int smooth) {
  if (!h || !path || !material18 || !trans16) return fail(-1, "NULL argument");
  EZ_TRY
  readObj(path, h->triangles, mat_from18(material18), mat4_from16(trans16), smooth != 0);
  return 0;
  EZ_CATCH
}
int ezrt_host_read_obj_text(EzrtHostScene* h, const char* text, int64_t len, const float material18[18],
                            const float trans16[16], int smooth) {
  if (!h || !text || len < 0 || !material18 || !trans16) return fail(-1, "NULL argument");
  EZ_TRY
  readObjText(text, (size_t)len, h->triangles, mat_from18(material18), mat4_from16(trans16), smooth != 0);
  return 0;
  EZ_CATCH
}
int ezrt_host_add_triangles(EzrtHostScene* h, const float* t, int n) {
  if (!h || !t || n < 0) return fail(-1, "NULL argument");
  EZ_TRY
  for (int i = 0; i < n; i++) {
    const float* p = t + (size_t)i * 36;
    Triangle tr;
    tr.p1 = vec3(p[0], p[1], p[2]);
    tr.p2 = vec3(p[3], p[4], p[5]);
    tr.p3 = vec3(p[6], p[7], p[8]);
    tr.n1 = vec3(p[9], p[10], p[11]);
    tr.n2 = vec3(p[12], p[13], p[14]);
    tr.n3 = vec3(p[15], p[16], p[17]);
    tr.material = mat_from18(p + 18);
    h->triangles.push_back(tr);
  }
  return 0;
  EZ_CATCH
}

int ezrt_host_build_bvh(EzrtHostScene* h, int method, int leaf_n) {
  if (!h) return fail(-1, "NULL argument");
  if (h->triangles.empty()) return fail(-1, "no triangles");
  if (method != 0 && method != 1) return fail(-1, "method must be 0 (median) or 1 (SAH)");
  EZ_TRY
  h->nodes.clear();
  h->nodes.push_back(testNode());
  int n = (int)h->triangles.size();
  if (method == 0) buildBVH(h->triangles, h->nodes, 0, n - 1, leaf_n);
  else buildBVHwithSAH(h->triangles, h->nodes, 0, n - 1, leaf_n);
  h->stats = lastBuildStats();
  return 0;
  EZ_CATCH
}
int ezrt_host_set_tie_order(int library_sort) {
  if (library_sort != 0 && library_sort != 1) return fail(-1, "tie order must be 0 (stable) or 1 (library std::sort)");
  setTieOrder(library_sort ? TieOrder::LibrarySort : TieOrder::Stable);
  return 0;
}
int ezrt_host_build_stats(EzrtHostScene* h, int64_t out[3]) {
  if (!h || !out) return fail(-1, "NULL argument");
  out[0] = h->stats.inf_cap_nodes;
  out[1] = h->stats.sorts;
  out[2] = h->stats.max_depth;
  return 0;
}
int ezrt_host_counts(EzrtHostScene* h, int* n_tri, int* n_nodes) {
  if (!h) return fail(-1, "NULL argument");
  if (n_tri) *n_tri = (int)h->triangles.size();
  if (n_nodes) *n_nodes = (int)h->nodes.size();
  return 0;
}
int ezrt_host_encode(EzrtHostScene* h, float* tri_out, float* nodes_out) {
  if (!h) return fail(-1, "NULL argument");
  EZ_TRY
  if (tri_out) {
    std::vector<Triangle_encoded> e = encodeTriangles(h->triangles);
    if (!e.empty()) memcpy(tri_out, e.data(), e.size() * sizeof(Triangle_encoded));
  }
  if (nodes_out) {
    std::vector<BVHNode_encoded> e = encodeBVH(h->nodes);
    if (!e.empty()) memcpy(nodes_out, e.data(), e.size() * sizeof(BVHNode_encoded));
  }
  return 0;
  EZ_CATCH
}

static int hand_over(HDRLoaderResult& r, bool ok, int* w, int* h, float** data) {
  if (!ok) {
    delete[] r.cols;
    return fail(-1, "HDRLoader::load failed (not a Radiance file or unreadable)");
  }
  size_t n = (size_t)r.width * r.height * 3;
  float* out = (float*)malloc(n * sizeof(float));
  if (!out) {
    delete[] r.cols;
    return fail(-4, "out of memory");
  }
  memcpy(out, r.cols, n * sizeof(float));
  delete[] r.cols;
  *w = r.width;
  *h = r.height;
  *data = out;
  return 0;
}
int ezrt_host_hdr_load(const char* path, int* w, int* h, float** data) {
  if (!path || !w || !h || !data) return fail(-1, "NULL argument");
  EZ_TRY
  HDRLoaderResult r;
  bool ok = HDRLoader::load(path, r);
  return hand_over(r, ok, w, h, data);
  EZ_CATCH
}
int ezrt_host_hdr_load_memory(const unsigned char* bytes, int64_t len, int* w, int* h, float** data) {
  if (!bytes || len < 0 || !w || !h || !data) return fail(-1, "NULL argument");
  EZ_TRY
  HDRLoaderResult r;
  bool ok = HDRLoader::loadMemory(bytes, (size_t)len, r);
  return hand_over(r, ok, w, h, data);
  EZ_CATCH
}
int ezrt_host_hdr_cache(const float* hdr, int w, int h, float* out) {
  if (!hdr || !out || w <= 0 || h <= 0) return fail(-1, "bad argument");
  EZ_TRY
  float* c = calculateHdrCache(hdr, w, h);
  memcpy(out, c, (size_t)w * h * 3 * sizeof(float));
  delete[] c;
  return 0;
  EZ_CATCH
}
void ezrt_host_free(void* p) { free(p); }

int ezrt_host_camera(float rot, float up, float r, float eye3[3], float cam16[16]) {
  if (!eye3 || !cam16) return fail(-1, "NULL argument");
  Camera c = cameraFromAngles(rot, up, r);
  eye3[0] = c.eye.x;
  eye3[1] = c.eye.y;
  eye3[2] = c.eye.z;
  mat4_to16(c.cameraRotate, cam16);
  return 0;
}

int ezrt_host_p2_query(const float* tri9, int n_tri, int method, int leaf_n, const float* rays6, int n_rays,
                       int use_bvh, float* tri_sorted9, int* hit_index, float* hit_t) {
  if (!tri9 || n_tri <= 0 || (n_rays > 0 && (!rays6 || !hit_index || !hit_t))) return fail(-1, "bad argument");
  if (method != 0 && method != 1) return fail(-1, "method must be 0 (median) or 1 (SAH)");
  EZ_TRY
  std::vector<p2::Triangle> tris;
  tris.reserve((size_t)n_tri);
  for (int i = 0; i < n_tri; i++) {
    const float* p = tri9 + (size_t)i * 9;
    tris.push_back(p2::Triangle(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]), vec3(p[6], p[7], p[8])));
  }
  p2::BVHNode* root = method ? p2::buildBVHwithSAH(tris, 0, n_tri - 1, leaf_n) : p2::buildBVH(tris, 0, n_tri - 1, leaf_n);
  if (tri_sorted9)
    for (int i = 0; i < n_tri; i++) {
      float* p = tri_sorted9 + (size_t)i * 9;
      const p2::Triangle& t = tris[(size_t)i];
      p[0] = t.p1.x; p[1] = t.p1.y; p[2] = t.p1.z;
      p[3] = t.p2.x; p[4] = t.p2.y; p[5] = t.p2.z;
      p[6] = t.p3.x; p[7] = t.p3.y; p[8] = t.p3.z;
    }
  for (int i = 0; i < n_rays; i++) {
    const float* q = rays6 + (size_t)i * 6;
    p2::Ray ray;
    ray.startPoint = vec3(q[0], q[1], q[2]);
    ray.direction = vec3(q[3], q[4], q[5]);
    p2::HitResult res = use_bvh ? p2::hitBVH(ray, tris, root) : p2::hitTriangleArray(ray, tris, 0, n_tri - 1);
    hit_index[i] = res.triangle ? (int)(res.triangle - tris.data()) : -1;
    hit_t[i] = res.distance;
  }
  p2::freeBVH(root);
  return 0;
  EZ_CATCH
}

const char* ezrt_host_last_error(void) { return g_err.c_str(); }

} // extern "C"
