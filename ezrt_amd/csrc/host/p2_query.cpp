// p2_query.cpp -- chapter 2's CPU query API on a pointer tree (include/ezrt_scene.hpp,
// namespace ezrt::p2).  The builders are the flat builders of scene.cpp run on the
// same geometry (P2/main.cpp:242-423 and P3/main.cpp:394-588 are the same
// algorithm with a different node container), re-linked into heap nodes; the
// intersectors follow the fp32 contract of DESIGN.md section 2.
#include <cmath>

#include "ezrt_scene.hpp"

namespace ezrt {
namespace p2 {

namespace {
inline float fmin2(float a, float b) { return (b < a) ? b : a; }
inline float fmax2(float a, float b) { return (a < b) ? b : a; }
inline vec3 sub(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 add(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 mulv(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 muls(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline float dot3(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross3(vec3 a, vec3 b) {
  return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline vec3 normalize3(vec3 a) { return muls(a, 1.0f / std::sqrt(dot3(a, a))); }

BVHNode* link(const std::vector<ezrt::BVHNode>& flat, int id) {
  if (id <= 0) return nullptr;
  const ezrt::BVHNode& f = flat[(size_t)id];
  BVHNode* node = new BVHNode();
  node->n = f.n;
  node->index = f.index;
  node->AA = f.AA;
  node->BB = f.BB;
  if (f.n <= 0) {
    node->left = link(flat, f.left);
    node->right = link(flat, f.right);
  }
  return node;
}

BVHNode* build(std::vector<Triangle>& triangles, int l, int r, int n, bool sah) {
  if (l > r || l < 0 || r >= (int)triangles.size()) return nullptr;
  std::vector<ezrt::Triangle> work((size_t)(r - l + 1));
  for (int i = l; i <= r; ++i) {
    work[(size_t)(i - l)].p1 = triangles[(size_t)i].p1;
    work[(size_t)(i - l)].p2 = triangles[(size_t)i].p2;
    work[(size_t)(i - l)].p3 = triangles[(size_t)i].p3;
  }
  std::vector<ezrt::BVHNode> flat;
  flat.push_back(testNode());
  int root = sah ? ezrt::buildBVHwithSAH(work, flat, 0, r - l, n) : ezrt::buildBVH(work, flat, 0, r - l, n);
  for (int i = l; i <= r; ++i) // the builders sorted `work` in place: carry the order over
    triangles[(size_t)i] = Triangle(work[(size_t)(i - l)].p1, work[(size_t)(i - l)].p2, work[(size_t)(i - l)].p3);
  if (l != 0)
    for (ezrt::BVHNode& f : flat)
      if (f.n > 0) f.index += l;
  return link(flat, root);
}
} // namespace

Triangle::Triangle(vec3 a, vec3 b, vec3 c) : p1(a), p2(b), p3(c) {
  vec3 s = add(add(a, b), c);
  center = vec3(s.x / 3.0f, s.y / 3.0f, s.z / 3.0f);
}

BVHNode* buildBVH(std::vector<Triangle>& triangles, int l, int r, int n) { return build(triangles, l, r, n, false); }
BVHNode* buildBVHwithSAH(std::vector<Triangle>& triangles, int l, int r, int n) {
  return build(triangles, l, r, n, true);
}

void freeBVH(BVHNode* root) {
  if (!root) return;
  freeBVH(root->left);
  freeBVH(root->right);
  delete root;
}

float hitTriangle(Triangle* triangle, Ray ray) {
  vec3 p1 = triangle->p1, p2 = triangle->p2, p3 = triangle->p3;
  vec3 S = ray.startPoint, d = ray.direction;
  vec3 N = normalize3(cross3(sub(p2, p1), sub(p3, p1)));
  if (dot3(N, d) > 0.0f) N = vec3(-N.x, -N.y, -N.z);
  if (std::fabs(dot3(N, d)) < 0.00001f) return INF;
  float t = (dot3(N, p1) - dot3(S, N)) / dot3(d, N);
  if (t < 0.0005f) return INF;
  vec3 P = add(S, muls(d, t));
  vec3 c1 = cross3(sub(p2, p1), sub(P, p1));
  vec3 c2 = cross3(sub(p3, p2), sub(P, p2));
  vec3 c3 = cross3(sub(p1, p3), sub(P, p3));
  float a = dot3(c1, N), b = dot3(c2, N), c = dot3(c3, N);
  if (a > 0 && b > 0 && c > 0) return t;
  if (a < 0 && b < 0 && c < 0) return t;
  return INF;
}

float hitAABB(Ray r, vec3 AA, vec3 BB) {
  vec3 invdir = vec3(1.0f / r.direction.x, 1.0f / r.direction.y, 1.0f / r.direction.z);
  vec3 in = mulv(sub(BB, r.startPoint), invdir);
  vec3 out = mulv(sub(AA, r.startPoint), invdir);
  vec3 tmax = vec3(fmax2(in.x, out.x), fmax2(in.y, out.y), fmax2(in.z, out.z));
  vec3 tmin = vec3(fmin2(in.x, out.x), fmin2(in.y, out.y), fmin2(in.z, out.z));
  float t1 = fmin2(tmax.x, fmin2(tmax.y, tmax.z));
  float t0 = fmax2(tmin.x, fmax2(tmin.y, tmin.z));
  return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

HitResult hitTriangleArray(Ray ray, std::vector<Triangle>& triangles, int l, int r) {
  HitResult res;
  for (int i = l; i <= r; i++) {
    float d = hitTriangle(&triangles[(size_t)i], ray);
    if (d < INF && d < res.distance) {
      res.distance = d;
      res.triangle = &triangles[(size_t)i];
    }
  }
  return res;
}

HitResult hitBVH(Ray ray, std::vector<Triangle>& triangles, BVHNode* root) {
  if (root == nullptr) return HitResult();
  if (root->n > 0) return hitTriangleArray(ray, triangles, root->index, root->index + root->n - 1);
  float d1 = INF, d2 = INF;
  if (root->left) d1 = hitAABB(ray, root->left->AA, root->left->BB);
  if (root->right) d2 = hitAABB(ray, root->right->AA, root->right->BB);
  HitResult r1, r2;
  if (d1 > 0) r1 = hitBVH(ray, triangles, root->left);
  if (d2 > 0) r2 = hitBVH(ray, triangles, root->right);
  return r1.distance < r2.distance ? r1 : r2;
}

} // namespace p2
} // namespace ezrt
