// scene.cpp -- host scene build: linear algebra PODs, OBJ reader, BVH builders,
// record encoding, camera.  New code restating the behaviour of the reference
// host (P3/main.cpp:254-588, 607-610, 720-748); see include/ezrt_scene.hpp for
// the interface map.  Build with -ffp-contract=off (results are inputs of the
// bit-exact trace parity tests).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "ezrt_detmath.h"
#include "ezrt_scene.hpp"

namespace ezrt {

// ---------------------------------------------------------------------------
// GLM-compatible helpers with a fixed evaluation order.

static inline float gmin(float a, float b) { return (b < a) ? b : a; } // glm::min
static inline float gmax(float a, float b) { return (a < b) ? b : a; } // glm::max
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) {
  return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline vec3 normalize(vec3 a) {
  float inv = 1.0f / __builtin_sqrtf(dot(a, a));
  return a * inv;
}

float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

mat4 identity() {
  mat4 m;
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) m.c[c][r] = (c == r) ? 1.0f : 0.0f;
  return m;
}
// column helpers
static inline void col_axpy(float out[4], const float a[4], float s) {
  for (int r = 0; r < 4; r++) out[r] = a[r] * s;
}
mat4 translate(const mat4& m, vec3 v) { // Result[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
  mat4 o = m;
  for (int r = 0; r < 4; r++) o.c[3][r] = ((m.c[0][r] * v.x + m.c[1][r] * v.y) + m.c[2][r] * v.z) + m.c[3][r];
  return o;
}
mat4 scale(const mat4& m, vec3 v) {
  mat4 o;
  col_axpy(o.c[0], m.c[0], v.x);
  col_axpy(o.c[1], m.c[1], v.y);
  col_axpy(o.c[2], m.c[2], v.z);
  for (int r = 0; r < 4; r++) o.c[3][r] = m.c[3][r];
  return o;
}
mat4 rotate(const mat4& m, float angle, vec3 v) {
  float s, c;
  ez_sincos(angle, &s, &c);
  vec3 axis = normalize(v);
  vec3 temp = axis * (1.0f - c);
  float R[3][3];
  R[0][0] = c + temp.x * axis.x;
  R[0][1] = temp.x * axis.y + s * axis.z;
  R[0][2] = temp.x * axis.z - s * axis.y;
  R[1][0] = temp.y * axis.x - s * axis.z;
  R[1][1] = c + temp.y * axis.y;
  R[1][2] = temp.y * axis.z + s * axis.x;
  R[2][0] = temp.z * axis.x + s * axis.y;
  R[2][1] = temp.z * axis.y - s * axis.x;
  R[2][2] = c + temp.z * axis.z;
  mat4 o;
  for (int j = 0; j < 3; j++)
    for (int r = 0; r < 4; r++) o.c[j][r] = (m.c[0][r] * R[j][0] + m.c[1][r] * R[j][1]) + m.c[2][r] * R[j][2];
  for (int r = 0; r < 4; r++) o.c[3][r] = m.c[3][r];
  return o;
}
mat4 mul(const mat4& a, const mat4& b) {
  mat4 o;
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 4; r++)
      o.c[j][r] = ((a.c[0][r] * b.c[j][0] + a.c[1][r] * b.c[j][1]) + a.c[2][r] * b.c[j][2]) + a.c[3][r] * b.c[j][3];
  return o;
}
vec4 mul(const mat4& m, vec4 v) {
  float o[4];
  for (int r = 0; r < 4; r++) o[r] = (m.c[0][r] * v.x + m.c[1][r] * v.y) + (m.c[2][r] * v.z + m.c[3][r] * v.w);
  return vec4{o[0], o[1], o[2], o[3]};
}
mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
  vec3 f = normalize(center - eye);
  vec3 s = normalize(cross(f, up));
  vec3 u = cross(s, f);
  mat4 o = identity();
  o.c[0][0] = s.x;
  o.c[1][0] = s.y;
  o.c[2][0] = s.z;
  o.c[0][1] = u.x;
  o.c[1][1] = u.y;
  o.c[2][1] = u.z;
  o.c[0][2] = -f.x;
  o.c[1][2] = -f.y;
  o.c[2][2] = -f.z;
  o.c[3][0] = -dot(s, eye);
  o.c[3][1] = -dot(u, eye);
  o.c[3][2] = dot(f, eye);
  return o;
}
mat4 inverse(const mat4& M) {
  // GLM's operand order (detail/func_matrix.inl, compute_inverse<4,4>): 2x2 sub-determinants shared
  // between cofactors, columns scaled by 1/det.  Bit-identical to the reference's inverse(lookAt(..))
  // on the same inputs, including the signs of zero entries (tests/test_ref_pin.py).
  const float(*m)[4] = M.c; // m[col][row]
  float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
  const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  const float v0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, v1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
  const float v2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, v3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
  mat4 inv;
  for (int k = 0; k < 4; k++) {
    const float sa = (k & 1) ? -1.0f : 1.0f, sb = -sa;
    inv.c[0][k] = ((v1[k] * f0[k] - v2[k] * f1[k]) + v3[k] * f2[k]) * sa;
    inv.c[1][k] = ((v0[k] * f0[k] - v2[k] * f3[k]) + v3[k] * f4[k]) * sb;
    inv.c[2][k] = ((v0[k] * f1[k] - v1[k] * f3[k]) + v3[k] * f5[k]) * sa;
    inv.c[3][k] = ((v0[k] * f2[k] - v1[k] * f4[k]) + v2[k] * f5[k]) * sb;
  }
  const float d0 = m[0][0] * inv.c[0][0], d1 = m[0][1] * inv.c[1][0], d2 = m[0][2] * inv.c[2][0], d3 = m[0][3] * inv.c[3][0];
  const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) inv.c[c][r] *= one_over_det;
  return inv;
}

Material disneyDefaults() {
  Material m;
  m.specular = 0.5f;
  m.roughness = 0.5f;
  m.sheenTint = 0.5f;
  m.clearcoatGloss = 1.0f;
  return m;
}

// P3/main.cpp:254-270: model = translate * (rotX * rotY * rotZ) * scale
mat4 getTransformMatrix(vec3 rotateCtrl, vec3 translateCtrl, vec3 scaleCtrl) {
  mat4 unit = identity();
  mat4 sc = scale(unit, scaleCtrl);
  mat4 tr = translate(unit, translateCtrl);
  mat4 rot = unit;
  rot = rotate(rot, radians(rotateCtrl.x), vec3(1, 0, 0));
  rot = rotate(rot, radians(rotateCtrl.y), vec3(0, 1, 0));
  rot = rotate(rot, radians(rotateCtrl.z), vec3(0, 0, 1));
  return mul(mul(tr, rot), sc);
}

// ---------------------------------------------------------------------------
// OBJ reader: P3/main.cpp:273-391.  Reproduced quirks: the extent
// bug (maxy/maxz/miny/minz are computed from maxx/minx), un-weighted vertex
// normal accumulation.  Face formats a, a/b, a/b/c chosen by slash count; as an
// extension "a//c" (which the reference's stream parse cannot read) is accepted.

static const char* skip_ws(const char* p, const char* e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
  return p;
}
static bool parse_float(const char*& p, const char* e, float& out) {
  p = skip_ws(p, e);
  if (p >= e) return false;
  char buf[64];
  size_t n = 0;
  while (p + n < e && n < sizeof(buf) - 1 && p[n] != ' ' && p[n] != '\t' && p[n] != '\r' && p[n] != '\n') n++;
  memcpy(buf, p, n);
  buf[n] = 0;
  char* end = nullptr;
  out = strtof(buf, &end);
  if (end == buf) return false;
  p += (end - buf);
  return true;
}
static bool parse_int(const char*& p, const char* e, int& out) {
  p = skip_ws(p, e);
  if (p >= e) return false;
  char buf[32];
  size_t n = 0;
  while (p + n < e && n < sizeof(buf) - 1 && (p[n] == '-' || p[n] == '+' || (p[n] >= '0' && p[n] <= '9'))) n++;
  if (n == 0) return false;
  memcpy(buf, p, n);
  buf[n] = 0;
  out = (int)strtol(buf, nullptr, 10);
  p += n;
  return true;
}
// one "v", "v/vt" or "v/vt/vn" (or "v//vn") group -> vertex index
static bool parse_face_vertex(const char*& p, const char* e, int& v) {
  if (!parse_int(p, e, v)) return false;
  while (p < e && *p == '/') {
    p++;
    int dummy;
    if (p < e && *p != '/' && *p != ' ' && *p != '\t' && *p != '\r') parse_int(p, e, dummy);
  }
  return true;
}

void readObjText(const char* text, size_t len, std::vector<Triangle>& triangles, Material material, mat4 trans,
                 bool smoothNormal) {
  std::vector<vec3> vertices;
  std::vector<unsigned> indices;
  float maxx = (float)-11451419.19, maxy = (float)-11451419.19, maxz = (float)-11451419.19;
  float minx = (float)11451419.19, miny = (float)11451419.19, minz = (float)11451419.19;
  const char* p = text;
  const char* end = text + len;
  while (p < end) {
    const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    const char* q = skip_ws(p, eol);
    // type token
    const char* t0 = q;
    while (q < eol && *q != ' ' && *q != '\t' && *q != '\r') q++;
    size_t tl = (size_t)(q - t0);
    if (tl == 1 && t0[0] == 'v') {
      float x = 0, y = 0, z = 0;
      parse_float(q, eol, x);
      parse_float(q, eol, y);
      parse_float(q, eol, z);
      vertices.push_back(vec3(x, y, z));
      // P3/main.cpp:316-317 -- sic: y/z extents are taken against maxx/minx
      maxx = gmax(maxx, x);
      maxy = gmax(maxx, y);
      maxz = gmax(maxx, z);
      minx = gmin(minx, x);
      miny = gmin(minx, y);
      minz = gmin(minx, z);
    } else if (tl == 1 && t0[0] == 'f') {
      int v0 = 0, v1 = 0, v2 = 0;
      if (parse_face_vertex(q, eol, v0) && parse_face_vertex(q, eol, v1) && parse_face_vertex(q, eol, v2)) {
        indices.push_back((unsigned)(v0 - 1));
        indices.push_back((unsigned)(v1 - 1));
        indices.push_back((unsigned)(v2 - 1));
      }
    }
    p = (eol < end) ? eol + 1 : end;
  }
  for (unsigned idx : indices)
    if (idx >= vertices.size()) throw std::runtime_error("readObj: face index outside the vertex list");

  float lenx = maxx - minx, leny = maxy - miny, lenz = maxz - minz;
  float maxaxis = gmax(lenx, gmax(leny, lenz));
  for (auto& v : vertices) {
    v.x /= maxaxis;
    v.y /= maxaxis;
    v.z /= maxaxis;
  }
  for (auto& v : vertices) {
    vec4 vv = mul(trans, vec4{v.x, v.y, v.z, 1.0f});
    v = vec3(vv.x, vv.y, vv.z);
  }
  std::vector<vec3> normals(vertices.size(), vec3(0, 0, 0));
  for (size_t i = 0; i + 2 < indices.size(); i += 3) {
    vec3 p1 = vertices[indices[i]], p2 = vertices[indices[i + 1]], p3 = vertices[indices[i + 2]];
    vec3 n = normalize(cross(p2 - p1, p3 - p1));
    normals[indices[i]] = normals[indices[i]] + n;
    normals[indices[i + 1]] = normals[indices[i + 1]] + n;
    normals[indices[i + 2]] = normals[indices[i + 2]] + n;
  }
  size_t offset = triangles.size();
  triangles.resize(offset + indices.size() / 3);
  for (size_t i = 0; i + 2 < indices.size(); i += 3) {
    Triangle& t = triangles[offset + i / 3];
    t.p1 = vertices[indices[i]];
    t.p2 = vertices[indices[i + 1]];
    t.p3 = vertices[indices[i + 2]];
    if (!smoothNormal) {
      vec3 n = normalize(cross(t.p2 - t.p1, t.p3 - t.p1));
      t.n1 = n;
      t.n2 = n;
      t.n3 = n;
    } else {
      t.n1 = normalize(normals[indices[i]]);
      t.n2 = normalize(normals[indices[i + 1]]);
      t.n3 = normalize(normals[indices[i + 2]]);
    }
    t.material = material;
  }
}

void readObj(const std::string& filepath, std::vector<Triangle>& triangles, Material material, mat4 trans,
             bool smoothNormal) {
  std::ifstream fin(filepath, std::ios::binary);
  if (!fin.is_open()) throw std::runtime_error("readObj: cannot open " + filepath);
  std::stringstream ss;
  ss << fin.rdbuf();
  std::string text = ss.str();
  readObjText(text.data(), text.size(), triangles, material, trans, smoothNormal);
}

// ---------------------------------------------------------------------------
// BVH builders: P3/main.cpp:394-588.
//
// The reference std::sorts 144-byte Triangle structs in place (4 sorts per SAH
// node) with comparators that recompute centroids.  std::sort's control flow
// depends only on comparison results, so sorting light (key, slot) records with
// the same std::sort and the same strict-< on the same key values yields the
// same permutation; triangles are physically permuted once at the end.  Bounds
// and centroids are pure functions of a triangle and are computed once.

static thread_local TieOrder g_tie_order = TieOrder::Stable;
void setTieOrder(TieOrder t) { g_tie_order = t; }
TieOrder tieOrder() { return g_tie_order; }

BVHNode testNode() { // P3/main.cpp:707-713 (index is left uninitialised there; 0 here)
  BVHNode n;
  n.left = 255;
  n.right = 128;
  n.n = 30;
  n.index = 0;
  n.AA = vec3(1, 1, 0);
  n.BB = vec3(0, 1, 0);
  return n;
}

namespace {

struct TriAux {
  float cen[3];
  float lo[3], hi[3];
};
struct SortRec {
  float key;
  int slot; // index into aux / original triangle subrange
};
struct Builder {
  std::vector<TriAux> aux;   // per original triangle of [l0, r0]
  std::vector<int> order;    // order[i - l0] = aux slot currently at array position i
  std::vector<SortRec> recs; // scratch
  std::vector<vec3> leftMax, leftMin, rightMax, rightMin;
  int l0 = 0;
  BuildStats stats;

  void init(const std::vector<Triangle>& tris, int l, int r) {
    l0 = l;
    int n = r - l + 1;
    aux.resize((size_t)n);
    order.resize((size_t)n);
    for (int i = 0; i < n; i++) {
      const Triangle& t = tris[(size_t)(l + i)];
      TriAux& a = aux[(size_t)i];
      // cmpx/cmpy/cmpz: center = (p1 + p2 + p3) / vec3(3,3,3)   P3/main.cpp:155-169
      a.cen[0] = ((t.p1.x + t.p2.x) + t.p3.x) / 3.0f;
      a.cen[1] = ((t.p1.y + t.p2.y) + t.p3.y) / 3.0f;
      a.cen[2] = ((t.p1.z + t.p2.z) + t.p3.z) / 3.0f;
      a.lo[0] = gmin(t.p1.x, gmin(t.p2.x, t.p3.x));
      a.lo[1] = gmin(t.p1.y, gmin(t.p2.y, t.p3.y));
      a.lo[2] = gmin(t.p1.z, gmin(t.p2.z, t.p3.z));
      a.hi[0] = gmax(t.p1.x, gmax(t.p2.x, t.p3.x));
      a.hi[1] = gmax(t.p1.y, gmax(t.p2.y, t.p3.y));
      a.hi[2] = gmax(t.p1.z, gmax(t.p2.z, t.p3.z));
      order[(size_t)i] = i;
    }
  }
  void sort_axis(int l, int r, int axis) { // std::sort(&tri[l], &tri[r]+1, cmp<axis>)
    int n = r - l + 1;
    recs.resize((size_t)n);
    for (int i = 0; i < n; i++) {
      int s = order[(size_t)(l - l0 + i)];
      recs[(size_t)i] = SortRec{aux[(size_t)s].cen[axis], s};
    }
    // std::sort leaves the order of equal keys to the library (parity unpinned); by default equal
    // keys keep their current order, which is what the numpy restatement and the GPU builder do too
    auto cmp = [](const SortRec& a, const SortRec& b) { return a.key < b.key; };
    if (g_tie_order == TieOrder::LibrarySort) std::sort(recs.begin(), recs.end(), cmp);
    else std::stable_sort(recs.begin(), recs.end(), cmp);
    for (int i = 0; i < n; i++) order[(size_t)(l - l0 + i)] = recs[(size_t)i].slot;
    stats.sorts++;
  }
  const TriAux& at(int i) const { return aux[(size_t)order[(size_t)(i - l0)]]; }

  int new_node(std::vector<BVHNode>& nodes, int l, int r) {
    nodes.push_back(BVHNode());
    int id = (int)nodes.size() - 1;
    BVHNode& nd = nodes[(size_t)id];
    nd.left = nd.right = nd.n = nd.index = 0;
    nd.AA = vec3((float)1145141919, (float)1145141919, (float)1145141919);
    nd.BB = vec3((float)-1145141919, (float)-1145141919, (float)-1145141919);
    for (int i = l; i <= r; i++) {
      const TriAux& a = at(i);
      nd.AA.x = gmin(nd.AA.x, a.lo[0]);
      nd.AA.y = gmin(nd.AA.y, a.lo[1]);
      nd.AA.z = gmin(nd.AA.z, a.lo[2]);
      nd.BB.x = gmax(nd.BB.x, a.hi[0]);
      nd.BB.y = gmax(nd.BB.y, a.hi[1]);
      nd.BB.z = gmax(nd.BB.z, a.hi[2]);
    }
    return id;
  }

  // P3/main.cpp:394-454
  int median(std::vector<BVHNode>& nodes, int l, int r, int n, int depth) {
    if (l > r) return 0;
    int id = new_node(nodes, l, r);
    if (depth > stats.max_depth) stats.max_depth = depth;
    if ((r - l + 1) <= n) {
      nodes[(size_t)id].n = r - l + 1;
      nodes[(size_t)id].index = l;
      return id;
    }
    float lenx = nodes[(size_t)id].BB.x - nodes[(size_t)id].AA.x;
    float leny = nodes[(size_t)id].BB.y - nodes[(size_t)id].AA.y;
    float lenz = nodes[(size_t)id].BB.z - nodes[(size_t)id].AA.z;
    // three independent ifs: on ties the last matching axis wins (Q7)
    if (lenx >= leny && lenx >= lenz) sort_axis(l, r, 0);
    if (leny >= lenx && leny >= lenz) sort_axis(l, r, 1);
    if (lenz >= lenx && lenz >= leny) sort_axis(l, r, 2);
    int mid = (l + r) / 2;
    int left = median(nodes, l, mid, n, depth + 1);
    int right = median(nodes, mid + 1, r, n, depth + 1);
    nodes[(size_t)id].left = left;
    nodes[(size_t)id].right = right;
    return id;
  }

  // P3/main.cpp:457-588
  int sah(std::vector<BVHNode>& nodes, int l, int r, int n, int depth) {
    if (l > r) return 0;
    int id = new_node(nodes, l, r);
    if (depth > stats.max_depth) stats.max_depth = depth;
    if ((r - l + 1) <= n) {
      nodes[(size_t)id].n = r - l + 1;
      nodes[(size_t)id].index = l;
      return id;
    }
    const float INF = 114514.0f;
    float Cost = INF;
    int Axis = 0;
    int Split = (l + r) / 2;
    int cnt = r - l + 1;
    leftMax.resize((size_t)cnt);
    leftMin.resize((size_t)cnt);
    rightMax.resize((size_t)cnt);
    rightMin.resize((size_t)cnt);
    for (int axis = 0; axis < 3; axis++) {
      sort_axis(l, r, axis);
      // prefix boxes; the running value starts at -+INF exactly as the reference's vectors do
      vec3 mx(-INF, -INF, -INF), mn(INF, INF, INF);
      for (int i = l; i <= r; i++) {
        const TriAux& a = at(i);
        mx.x = gmax(mx.x, a.hi[0]);
        mx.y = gmax(mx.y, a.hi[1]);
        mx.z = gmax(mx.z, a.hi[2]);
        mn.x = gmin(mn.x, a.lo[0]);
        mn.y = gmin(mn.y, a.lo[1]);
        mn.z = gmin(mn.z, a.lo[2]);
        leftMax[(size_t)(i - l)] = mx;
        leftMin[(size_t)(i - l)] = mn;
      }
      mx = vec3(-INF, -INF, -INF);
      mn = vec3(INF, INF, INF);
      for (int i = r; i >= l; i--) {
        const TriAux& a = at(i);
        mx.x = gmax(mx.x, a.hi[0]);
        mx.y = gmax(mx.y, a.hi[1]);
        mx.z = gmax(mx.z, a.hi[2]);
        mn.x = gmin(mn.x, a.lo[0]);
        mn.y = gmin(mn.y, a.lo[1]);
        mn.z = gmin(mn.z, a.lo[2]);
        rightMax[(size_t)(i - l)] = mx;
        rightMin[(size_t)(i - l)] = mn;
      }
      float cost = INF;
      int split = l;
      for (int i = l; i <= r - 1; i++) {
        vec3 lA = leftMin[(size_t)(i - l)], lB = leftMax[(size_t)(i - l)];
        float lenx = lB.x - lA.x, leny = lB.y - lA.y, lenz = lB.z - lA.z;
        float leftS = 2.0f * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
        float leftCost = leftS * (float)(i - l + 1);
        vec3 rA = rightMin[(size_t)(i + 1 - l)], rB = rightMax[(size_t)(i + 1 - l)];
        lenx = rB.x - rA.x;
        leny = rB.y - rA.y;
        lenz = rB.z - rA.z;
        float rightS = 2.0f * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
        float rightCost = rightS * (float)(r - i);
        float totalCost = leftCost + rightCost;
        if (totalCost < cost) {
          cost = totalCost;
          split = i;
        }
      }
      if (cost < Cost) {
        Cost = cost;
        Axis = axis;
        Split = split;
      }
    }
    if (!(Cost < INF)) stats.inf_cap_nodes++; // every candidate >= INF: median-x fallback (Q7)
    sort_axis(l, r, Axis);
    int left = sah(nodes, l, Split, n, depth + 1);
    int right = sah(nodes, Split + 1, r, n, depth + 1);
    nodes[(size_t)id].left = left;
    nodes[(size_t)id].right = right;
    return id;
  }

  void apply(std::vector<Triangle>& tris, int l, int r) {
    int n = r - l + 1;
    std::vector<Triangle> tmp((size_t)n);
    for (int i = 0; i < n; i++) tmp[(size_t)i] = tris[(size_t)(l + order[(size_t)i])];
    for (int i = 0; i < n; i++) tris[(size_t)(l + i)] = tmp[(size_t)i];
  }
};

thread_local BuildStats g_last_stats;

} // namespace

BuildStats lastBuildStats() { return g_last_stats; }

int buildBVH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n) {
  if (l > r) return 0;
  if (l < 0 || r >= (int)triangles.size()) throw std::out_of_range("buildBVH: [l, r] outside the triangle array");
  if (n < 1) throw std::invalid_argument("buildBVH: leaf size must be >= 1");
  Builder b;
  b.init(triangles, l, r);
  int id = b.median(nodes, l, r, n, 1);
  b.apply(triangles, l, r);
  g_last_stats = b.stats;
  return id;
}

int buildBVHwithSAH(std::vector<Triangle>& triangles, std::vector<BVHNode>& nodes, int l, int r, int n) {
  if (l > r) return 0;
  if (l < 0 || r >= (int)triangles.size())
    throw std::out_of_range("buildBVHwithSAH: [l, r] outside the triangle array");
  if (n < 1) throw std::invalid_argument("buildBVHwithSAH: leaf size must be >= 1");
  Builder b;
  b.init(triangles, l, r);
  int id = b.sah(nodes, l, r, n, 1);
  b.apply(triangles, l, r);
  g_last_stats = b.stats;
  return id;
}

// ---------------------------------------------------------------------------
// encode loops: P3/main.cpp:720-748

Triangle_encoded encodeTriangle(const Triangle& t) {
  const Material& m = t.material;
  Triangle_encoded e;
  e.p1 = t.p1;
  e.p2 = t.p2;
  e.p3 = t.p3;
  e.n1 = t.n1;
  e.n2 = t.n2;
  e.n3 = t.n3;
  e.emissive = m.emissive;
  e.baseColor = m.baseColor;
  e.param1 = vec3(m.subsurface, m.metallic, m.specular);
  e.param2 = vec3(m.specularTint, m.roughness, m.anisotropic);
  e.param3 = vec3(m.sheen, m.sheenTint, m.clearcoat);
  e.param4 = vec3(m.clearcoatGloss, m.IOR, m.transmission);
  return e;
}
BVHNode_encoded encodeBVH(const BVHNode& n) {
  BVHNode_encoded e;
  e.childs = vec3((float)n.left, (float)n.right, 0.0f);
  e.leafInfo = vec3((float)n.n, (float)n.index, 0.0f);
  e.AA = n.AA;
  e.BB = n.BB;
  return e;
}
std::vector<Triangle_encoded> encodeTriangles(const std::vector<Triangle>& triangles) {
  std::vector<Triangle_encoded> out(triangles.size());
  for (size_t i = 0; i < triangles.size(); i++) out[i] = encodeTriangle(triangles[i]);
  return out;
}
std::vector<BVHNode_encoded> encodeBVH(const std::vector<BVHNode>& nodes) {
  std::vector<BVHNode_encoded> out(nodes.size());
  for (size_t i = 0; i < nodes.size(); i++) out[i] = encodeBVH(nodes[i]);
  return out;
}

// ---------------------------------------------------------------------------
// camera: P3/main.cpp:607-610

Camera cameraFromAngles(float rotatAngle, float upAngle, float r) {
  float sr, cr, su, cu;
  ez_sincos(radians(rotatAngle), &sr, &cr);
  ez_sincos(radians(upAngle), &su, &cu);
  vec3 eye(-sr * cu, su, cr * cu);
  eye.x *= r;
  eye.y *= r;
  eye.z *= r;
  Camera c;
  c.eye = eye;
  c.cameraRotate = inverse(lookAt(eye, vec3(0, 0, 0), vec3(0, 1, 0)));
  return c;
}

} // namespace ezrt
