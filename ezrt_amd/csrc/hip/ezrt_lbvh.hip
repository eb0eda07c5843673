// ezrt_lbvh.hip -- GPU BVH builder: an ALTERNATIVE to the host buildBVHwithSAH
// (P3/main.cpp:457-588, O(n log^2 n) on 144-byte records) for scenes where the host build dominates.
// It produces the reference's data formats -- the triangle array reordered so that every leaf is a
// contiguous range, and 48-byte node records (node 0 dummy, root 1, child ids > parent id, leaves
// of at most leaf_n triangles) -- so its output feeds ezrt_scene_create, the CPU oracle and the
// reference's own shader alike.  It is NOT the parity default: node numbering and tree shape differ
// from the reference builder's; its contract is the reference's acceptance check "BVH == brute
// force" (P2/main.cpp:581-588) plus GPU == oracle on the same arrays (tests/test_gpu_lbvh.py).
//
// Algorithm (Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d
// trees"): 30-bit Morton codes of the triangle centroids, made unique by the triangle index
// (64-bit keys), radix sort (rocPRIM), one thread per internal node finds its key range and split
// by binary search on common-prefix lengths; subtrees of <= leaf_n triangles collapse into leaves;
// boxes and subtree sizes bottom-up behind per-node arrival counters; pre-order ids from the
// subtree sizes (left child = id + 1, right child = id + 1 + size(left)).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "ezrt.h"
#include "ezrt_build.h"

extern "C" int ezrt_fail_msg(int code, const char* msg); // ezrt_hip.hip: sets ezrt_last_error()

namespace {

#define LB_TRY(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      char buf[256];                                                              \
      snprintf(buf, sizeof buf, "%s failed: %s", #expr, hipGetErrorString(e_)); \
      return ezrt_fail_msg(EZRT_ERR_DEVICE, buf);                                 \
    }                                                                             \
  } while (0)

template <class T>
struct Buf {
  T* p = nullptr;
  ~Buf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

constexpr int TPB = 256;
constexpr int TRI_F = EZRT_TRI_FLOATS; // 36

// order-preserving float <-> uint map for atomicMin/atomicMax
__device__ inline uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ inline void tri_centroid(const float* t, float c[3]) {
  for (int k = 0; k < 3; k++) c[k] = (t[k] + t[3 + k] + t[6 + k]) / 3.0f; // as the reference's comparators
}

// bounds of all centroids: ord[0..2] = min, ord[3..5] = max
__global__ void k_centroid_bounds(const float* tri, int n, uint32_t* ord) {
  __shared__ uint32_t smin[3], smax[3];
  if (threadIdx.x < 3) {
    smin[threadIdx.x] = 0xffffffffu;
    smax[threadIdx.x] = 0u;
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float c[3];
    tri_centroid(tri + (size_t)i * TRI_F, c);
    for (int k = 0; k < 3; k++) {
      atomicMin(&smin[k], f2ord(c[k]));
      atomicMax(&smax[k], f2ord(c[k]));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    atomicMin(&ord[threadIdx.x], smin[threadIdx.x]);
    atomicMax(&ord[3 + threadIdx.x], smax[threadIdx.x]);
  }
}

__device__ inline uint32_t expand10(uint32_t v) { // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void k_morton(const float* tri, int n, const uint32_t* ord, unsigned long long* keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float c[3];
  tri_centroid(tri + (size_t)i * TRI_F, c);
  uint32_t q[3];
  for (int k = 0; k < 3; k++) {
    const float lo = ord2f(ord[k]), hi = ord2f(ord[3 + k]);
    const float ext = hi - lo;
    float u = ext > 0.0f ? (c[k] - lo) / ext : 0.0f;
    u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    uint32_t v = (uint32_t)(u * 1023.0f);
    q[k] = v > 1023u ? 1023u : v;
  }
  const uint32_t code = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
  keys[i] = ((unsigned long long)code << 32) | (uint32_t)i;
}

// Node space of the binary radix tree over n sorted keys: internal nodes [0, n-1), key-leaves
// [n-1, 2n-1) (key j = node n-1+j).
struct Tree {
  int n;
  const unsigned long long* keys;
  int* first;  // [n-1] range of an internal node
  int* last;   // [n-1]
  int* left;   // [n-1] child node ids
  int* right;  // [n-1]
  int* parent; // [2n-1]
};

__device__ inline int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j])); // keys are unique
}

__global__ void k_hierarchy(Tree t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = t.n;
  if (i >= n - 1) return;
  const unsigned long long* keys = t.keys;
  // direction of the range, its length by exponential + binary search (Karras, fig. 4)
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int step = lmax / 2; step >= 1; step /= 2)
    if (delta(keys, n, i, i + (l + step) * d) > dmin) l += step;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int step = (l + 1) / 2;; step = (step + 1) / 2) {
    if (delta(keys, n, i, i + (s + step) * d) > dnode) s += step;
    if (step <= 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int lc = (lo == gamma) ? (n - 1 + gamma) : gamma;
  const int rc = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
  t.first[i] = lo;
  t.last[i] = hi;
  t.left[i] = lc;
  t.right[i] = rc;
  t.parent[lc] = i;
  t.parent[rc] = i;
  if (i == 0) t.parent[0] = -1;
}

struct Fit {
  Tree t;
  int leaf_n;
  const float* tri;                 // input triangles
  const unsigned long long* keys;   // sorted: low word = input index
  float* box;                       // [2n-1][6]
  int* size;                        // [2n-1] kept nodes in the subtree (valid for kept nodes)
  uint32_t* arrived;                // [n-1]
};

__device__ inline int node_count(const Tree& t, int v) { return v >= t.n - 1 ? 1 : (t.last[v] - t.first[v] + 1); }
// a node is an output LEAF if it holds <= leaf_n triangles while its parent holds more (or it is the root)
__device__ inline bool is_out_leaf(const Tree& t, int leaf_n, int v) {
  if (node_count(t, v) > leaf_n) return false;
  const int p = t.parent[v];
  return p < 0 || node_count(t, p) > leaf_n;
}

// one thread per node of the radix tree: output leaves compute their box and climb
__global__ void k_fit(Fit f) {
  const int v0 = blockIdx.x * blockDim.x + threadIdx.x;
  const Tree& t = f.t;
  const int n = t.n;
  if (v0 >= 2 * n - 1) return;
  if (!is_out_leaf(t, f.leaf_n, v0)) return;
  int lo, hi;
  if (v0 >= n - 1) lo = hi = v0 - (n - 1);
  else {
    lo = t.first[v0];
    hi = t.last[v0];
  }
  float b[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
  for (int k = lo; k <= hi; k++) {
    const float* p = f.tri + (size_t)(uint32_t)f.keys[k] * TRI_F;
    for (int c = 0; c < 3; c++) {
      const float mn = fminf(p[c], fminf(p[3 + c], p[6 + c])), mx = fmaxf(p[c], fmaxf(p[3 + c], p[6 + c]));
      b[c] = fminf(b[c], mn);
      b[3 + c] = fmaxf(b[3 + c], mx);
    }
  }
  for (int c = 0; c < 6; c++) f.box[(size_t)v0 * 6 + c] = b[c];
  f.size[v0] = 1;
  __threadfence();
  int v = t.parent[v0];
  while (v >= 0) {
    if (atomicAdd(&f.arrived[v], 1u) == 0u) return; // the sibling subtree is not done yet
    __threadfence();
    const int l = t.left[v], r = t.right[v];
    for (int c = 0; c < 3; c++) {
      f.box[(size_t)v * 6 + c] = fminf(f.box[(size_t)l * 6 + c], f.box[(size_t)r * 6 + c]);
      f.box[(size_t)v * 6 + 3 + c] = fmaxf(f.box[(size_t)l * 6 + 3 + c], f.box[(size_t)r * 6 + 3 + c]);
    }
    f.size[v] = 1 + f.size[l] + f.size[r];
    __threadfence();
    v = t.parent[v];
  }
}

// one thread per node: kept nodes compute their pre-order id by walking up and emit their record
__global__ void k_emit(Fit f, float* nodes_out) {
  const int v0 = blockIdx.x * blockDim.x + threadIdx.x;
  const Tree& t = f.t;
  const int n = t.n;
  if (v0 >= 2 * n - 1) return;
  const bool leaf = is_out_leaf(t, f.leaf_n, v0);
  if (!leaf && node_count(t, v0) <= f.leaf_n) return; // swallowed by a leaf above it
  int id = 1; // root
  for (int v = v0, p = t.parent[v0]; p >= 0; v = p, p = t.parent[p])
    id += 1 + (t.right[p] == v ? f.size[t.left[p]] : 0);
  float* o = nodes_out + (size_t)id * EZRT_NODE_FLOATS;
  if (leaf) {
    const int lo = v0 >= n - 1 ? v0 - (n - 1) : t.first[v0];
    o[0] = 0.0f;
    o[1] = 0.0f;
    o[3] = (float)node_count(t, v0);
    o[4] = (float)lo;
  } else {
    o[0] = (float)(id + 1);
    o[1] = (float)(id + 1 + f.size[t.left[v0]]);
    o[3] = 0.0f;
    o[4] = 0.0f;
  }
  o[2] = 0.0f;
  o[5] = 0.0f;
  for (int c = 0; c < 6; c++) o[6 + c] = f.box[(size_t)v0 * 6 + c];
}

__global__ void k_gather(const float4* tri_in, const unsigned long long* keys, int n, float4* tri_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one float4 (of 9) per thread
  if (i >= (size_t)n * 9) return;
  const size_t t = i / 9, q = i % 9;
  tri_out[i] = tri_in[(size_t)(uint32_t)keys[t] * 9 + q];
}

} // namespace

extern "C" int ezrt_build_lbvh(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out,
                               int nodes_capacity, int* n_nodes, float* build_ms) {
  if (!tri || !tri_out || !nodes_out || !n_nodes) return ezrt_fail_msg(EZRT_ERR_INVALID, "NULL argument");
  if (n_tri <= 0) return ezrt_fail_msg(EZRT_ERR_INVALID, "no triangles");
  if (n_tri >= (1 << 24)) return ezrt_fail_msg(EZRT_ERR_UNSUPPORTED, "counts >= 2^24 are not exact in the float encoding");
  if (leaf_n < 1 || leaf_n > 128) return ezrt_fail_msg(EZRT_ERR_INVALID, "leaf_n must be in [1, 128]");
  const int n = n_tri;
  const size_t tri_bytes = (size_t)n * TRI_F * sizeof(float);
  Buf<float> d_tri, d_tri_out, d_nodes, d_box;
  Buf<uint32_t> d_ord, d_arrived;
  Buf<unsigned long long> d_keys, d_keys2;
  Buf<int> d_first, d_last, d_left, d_right, d_parent, d_size;
  Buf<char> d_tmp;
  LB_TRY(d_tri.alloc((size_t)n * TRI_F));
  LB_TRY(d_tri_out.alloc((size_t)n * TRI_F));
  LB_TRY(d_ord.alloc(6));
  LB_TRY(d_keys.alloc(n));
  LB_TRY(d_keys2.alloc(n));
  LB_TRY(d_first.alloc(n));
  LB_TRY(d_last.alloc(n));
  LB_TRY(d_left.alloc(n));
  LB_TRY(d_right.alloc(n));
  LB_TRY(d_parent.alloc(2 * (size_t)n));
  LB_TRY(d_size.alloc(2 * (size_t)n));
  LB_TRY(d_box.alloc(2 * (size_t)n * 6));
  LB_TRY(d_arrived.alloc(n));
  LB_TRY(d_nodes.alloc(2 * (size_t)n * EZRT_NODE_FLOATS));
  size_t tmp_bytes = 0;
  LB_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.p, d_keys2.p, (size_t)n, 0, 62, nullptr));
  LB_TRY(d_tmp.alloc(tmp_bytes));
  LB_TRY(hipMemcpy(d_tri.p, tri, tri_bytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  LB_TRY(hipEventCreate(&e0));
  LB_TRY(hipEventCreate(&e1));
  LB_TRY(hipEventRecord(e0, nullptr));

  const uint32_t ord_init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LB_TRY(hipMemcpyAsync(d_ord.p, ord_init, sizeof ord_init, hipMemcpyHostToDevice, nullptr));
  const int grid_n = (n + TPB - 1) / TPB;
  hipLaunchKernelGGL(k_centroid_bounds, dim3(grid_n < 1024 ? grid_n : 1024), dim3(TPB), 0, nullptr, d_tri.p, n, d_ord.p);
  hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(TPB), 0, nullptr, d_tri.p, n, d_ord.p, d_keys.p);
  LB_TRY(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, (size_t)n, 0, 62, nullptr));

  int total_nodes = 2; // dummy + root
  LB_TRY(hipMemsetAsync(d_parent.p, 0xff, 2 * (size_t)n * sizeof(int), nullptr));
  Tree t;
  t.n = n;
  t.keys = d_keys2.p;
  t.first = d_first.p;
  t.last = d_last.p;
  t.left = d_left.p;
  t.right = d_right.p;
  t.parent = d_parent.p;
  Fit f;
  f.t = t;
  f.leaf_n = leaf_n;
  f.tri = d_tri.p;
  f.keys = d_keys2.p;
  f.box = d_box.p;
  f.size = d_size.p;
  f.arrived = d_arrived.p;
  if (n > 1) hipLaunchKernelGGL(k_hierarchy, dim3((n - 1 + TPB - 1) / TPB), dim3(TPB), 0, nullptr, t);
  LB_TRY(hipMemsetAsync(d_arrived.p, 0, (size_t)n * sizeof(uint32_t), nullptr));
  const int grid_2n = (2 * n - 1 + TPB - 1) / TPB;
  hipLaunchKernelGGL(k_fit, dim3(grid_2n), dim3(TPB), 0, nullptr, f);
  hipLaunchKernelGGL(k_emit, dim3(grid_2n), dim3(TPB), 0, nullptr, f, d_nodes.p);
  hipLaunchKernelGGL(k_gather, dim3((unsigned)(((size_t)n * 9 + TPB - 1) / TPB)), dim3(TPB), 0, nullptr,
                     reinterpret_cast<const float4*>(d_tri.p), d_keys2.p, n, reinterpret_cast<float4*>(d_tri_out.p));
  LB_TRY(hipEventRecord(e1, nullptr));
  LB_TRY(hipEventSynchronize(e1));
  LB_TRY(hipGetLastError());
  float ms = 0.0f;
  LB_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  int root_size = 0; // the root is node 0 of the radix tree (for n == 1 the only key-leaf is node 0 too)
  LB_TRY(hipMemcpy(&root_size, d_size.p, sizeof(int), hipMemcpyDeviceToHost));
  total_nodes = 1 + root_size;
  if (total_nodes > nodes_capacity) {
    char buf[128];
    snprintf(buf, sizeof buf, "nodes_capacity %d too small: the tree has %d nodes", nodes_capacity, total_nodes);
    return ezrt_fail_msg(EZRT_ERR_INVALID, buf);
  }
  LB_TRY(hipMemcpy(tri_out, d_tri_out.p, tri_bytes, hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(nodes_out, d_nodes.p, (size_t)total_nodes * EZRT_NODE_FLOATS * sizeof(float), hipMemcpyDeviceToHost));
  // node 0: the reference's testNode sentinel (P3/main.cpp:707-713), as ezrt::testNode() encodes it
  const float sentinel[EZRT_NODE_FLOATS] = {255.0f, 128.0f, 0.0f, 30.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f};
  for (int k = 0; k < EZRT_NODE_FLOATS; k++) nodes_out[k] = sentinel[k];
  *n_nodes = total_nodes;
  if (build_ms) *build_ms = ms;
  return 0;
}
