// ezrt_sahbvh.hip -- buildBVHwithSAH (P3/main.cpp:457-588) on the GPU, producing EXACTLY the arrays of
// the host builder (ezrt::buildBVHwithSAH + the encode loops): the same triangle order, the same
// nodes with the same ids.  The reference's algorithm is kept as it is -- per node: sort by centroid
// on x, y, z in turn, prefix/suffix boxes, cost(i) = S_left (i-l+1) + S_right (r-i), best (axis,
// split) by strict <, the INF = 114514 cap with its median-x fallback, a last sort on the best axis,
// recursion on [l, Split] and [Split+1, r], leaves of <= n triangles -- only the schedule changes:
// all nodes of one tree level are processed together, so one level costs four stable radix sorts,
// seven segmented scans and three segmented arg-mins over the whole triangle array instead of
// O(nodes) std::sort calls.  O(n log n) per level on the device; the host keeps the node list (a few
// hundred thousand records) and numbers the nodes in the reference's creation order at the end.
//
// Why it is bit-identical: equal centroid keys keep their current order in both builders (stable
// sorts: std::stable_sort on the host, LSD radix sort here); min/max are exact and the scans keep
// the operand order (first-minimum semantics, so even +-0 ties agree); the cost expression is
// evaluated in the same fp32 order (-ffp-contract=off on both sides).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_reduce.hpp>

#include "ezrt.h"
#include "ezrt_build.h"

extern "C" int ezrt_fail_msg(int code, const char* msg); // ezrt_hip.hip: sets ezrt_last_error()

namespace {

#define SB_TRY(expr)                                                              \
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
  size_t n = 0;
  ~Buf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t count) {
    n = count ? count : 1;
    return hipMalloc((void**)&p, n * sizeof(T));
  }
};

constexpr int TPB = 256;
constexpr int TRI_F = EZRT_TRI_FLOATS;
constexpr float SAH_INF = 114514.0f;

__host__ __device__ inline float gmin(float a, float b) { return (b < a) ? b : a; } // glm::min
__host__ __device__ inline float gmax(float a, float b) { return (a < b) ? b : a; } // glm::max

struct Box {
  float lo[3], hi[3];
};
struct BoxOp { // running min/max in operand order (first minimum / first maximum wins)
  __host__ __device__ Box operator()(const Box& a, const Box& b) const {
    Box r;
    for (int k = 0; k < 3; k++) {
      r.lo[k] = gmin(a.lo[k], b.lo[k]);
      r.hi[k] = gmax(a.hi[k], b.hi[k]);
    }
    return r;
  }
};
struct U64Min {
  __host__ __device__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return b < a ? b : a; }
};

struct Aux { // per input triangle
  float cen[3];
  Box box;
};

__global__ void k_aux(const float* tri, int n, Aux* aux, int* order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* t = tri + (size_t)i * TRI_F;
  Aux a;
  for (int k = 0; k < 3; k++) {
    a.cen[k] = ((t[k] + t[3 + k]) + t[6 + k]) / 3.0f; // cmpx/cmpy/cmpz, P3/main.cpp:155-169
    a.box.lo[k] = gmin(t[k], gmin(t[3 + k], t[6 + k]));
    a.box.hi[k] = gmax(t[k], gmax(t[3 + k], t[6 + k]));
  }
  aux[i] = a;
  order[i] = i;
}

// frontier entry of the current level: a range of the triangle array
struct Seg {
  int l, r;
  int split;  // 1: more than leaf_n triangles, to be split at this level
  int axis;   // sort axis for the level's last sort (filled after the three sweeps)
};

__global__ void k_heads(const Seg* segs, int n_segs, int* head) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_segs) head[segs[k].l] = 1;
}

__device__ inline uint32_t ord_f32(float f) { // order-preserving; -0 and +0 compare equal in `<`, so fold them
  f = f + 0.0f;
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// sort key of position i: (start of its range, centroid on `axis`) inside ranges that are being split,
// (start, offset) elsewhere -- a stable sort then permutes only inside those ranges.  axis < 0: take
// the range's own axis (the last sort of the level).
__global__ void k_keys(const Seg* segs, const int* seg_of, const int* order, const Aux* aux, int n, int axis,
                       unsigned long long* keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Seg s = segs[seg_of[i] - 1];
  uint32_t lo;
  if (s.split) lo = ord_f32(aux[order[i]].cen[axis < 0 ? s.axis : axis]);
  else lo = (uint32_t)(i - s.l);
  keys[i] = ((unsigned long long)(uint32_t)s.l << 32) | lo;
}

__global__ void k_gather_boxes(const int* order, const Aux* aux, const int* seg_of, const Seg* segs, int n, Box* fwd,
                               uint32_t* fwd_key, Box* rev, uint32_t* rev_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Box b = aux[order[i]].box;
  const uint32_t key = (uint32_t)segs[seg_of[i] - 1].l;
  fwd[i] = b;
  fwd_key[i] = key;
  rev[n - 1 - i] = b;
  rev_key[n - 1 - i] = key;
}

// cost of splitting after position i (P3/main.cpp:540-566), as a sortable key: cost bits << 32 | i - l
__global__ void k_cost(const Seg* segs, const int* seg_of, const Box* pre, const Box* suf_rev, int n,
                       unsigned long long* cost_key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Seg s = segs[seg_of[i] - 1];
  unsigned long long key = ~0ull;
  if (s.split && i < s.r) {
    const Box L = pre[i], R = suf_rev[n - 1 - (i + 1)];
    float lenx = gmax(-SAH_INF, L.hi[0]) - gmin(SAH_INF, L.lo[0]);
    float leny = gmax(-SAH_INF, L.hi[1]) - gmin(SAH_INF, L.lo[1]);
    float lenz = gmax(-SAH_INF, L.hi[2]) - gmin(SAH_INF, L.lo[2]);
    const float leftS = 2.0f * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
    const float leftCost = leftS * (float)(i - s.l + 1);
    lenx = gmax(-SAH_INF, R.hi[0]) - gmin(SAH_INF, R.lo[0]);
    leny = gmax(-SAH_INF, R.hi[1]) - gmin(SAH_INF, R.lo[1]);
    lenz = gmax(-SAH_INF, R.hi[2]) - gmin(SAH_INF, R.lo[2]);
    const float rightS = 2.0f * ((lenx * leny) + (lenx * lenz) + (leny * lenz));
    const float rightCost = rightS * (float)(s.r - i);
    const float total = leftCost + rightCost;
    if (total < SAH_INF) // (only such candidates can ever satisfy totalCost < cost; costs are >= +0)
      key = ((unsigned long long)__float_as_uint(total) << 32) | (uint32_t)(i - s.l);
  }
  cost_key[i] = key;
}

struct Best {
  float cost;
  int axis, split;
};
__global__ void k_best(const unsigned long long* seg_min, const int* split_ids, const Seg* segs, int n_split, int axis,
                       Best* best) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_split) return;
  const Seg s = segs[split_ids[k]];
  Best b = best[k];
  if (axis == 0) {
    b.cost = SAH_INF;
    b.axis = 0;
    b.split = (s.l + s.r) / 2;
  }
  const unsigned long long m = seg_min[k];
  if (m != ~0ull) {
    const float cost = __uint_as_float((uint32_t)(m >> 32));
    if (cost < b.cost) {
      b.cost = cost;
      b.axis = axis;
      b.split = s.l + (int)(uint32_t)m;
    }
  }
  best[k] = b;
}
__global__ void k_set_axis(const int* split_ids, const Best* best, int n_split, Seg* segs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_split) segs[split_ids[k]].axis = best[k].axis;
}

__global__ void k_pick_boxes(const Box* pre, const int* pos, int m, Box* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) out[k] = pre[pos[k]];
}

__global__ void k_gather_tris(const float4* tri_in, const int* order, int n, float4* tri_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * 9) return;
  tri_out[i] = tri_in[(size_t)order[i / 9] * 9 + i % 9];
}

struct HostNode { // the reference's BVHNode + what the numbering pass needs
  int left = 0, right = 0, n = 0, index = 0;
  float AA[3], BB[3];
  int l = 0, r = 0;
};

} // namespace

namespace {
// median = false: buildBVHwithSAH (P3/main.cpp:457-588); true: buildBVH (P3/main.cpp:394-454)
int build_level_sync(bool median, const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out,
                     int nodes_capacity, int* n_nodes, float* build_ms) {
  if (!tri || !tri_out || !nodes_out || !n_nodes) return ezrt_fail_msg(EZRT_ERR_INVALID, "NULL argument");
  if (n_tri <= 0) return ezrt_fail_msg(EZRT_ERR_INVALID, "no triangles");
  if (n_tri >= (1 << 24)) return ezrt_fail_msg(EZRT_ERR_UNSUPPORTED, "counts >= 2^24 are not exact in the float encoding");
  if (leaf_n < 1) return ezrt_fail_msg(EZRT_ERR_INVALID, "leaf_n must be positive");
  const int n = n_tri;
  const int grid_n = (n + TPB - 1) / TPB;
  Buf<float> d_tri, d_tri_out;
  Buf<Aux> d_aux;
  Buf<int> d_order[2], d_head, d_seg_of, d_split_ids, d_off_b, d_off_e;
  Buf<unsigned long long> d_keys[2], d_cost, d_seg_min;
  Buf<Box> d_fwd, d_rev, d_pre, d_suf, d_node_pre;
  Buf<uint32_t> d_fwd_key, d_rev_key;
  Buf<Seg> d_segs;
  Buf<Best> d_best;
  Buf<char> d_tmp;
  SB_TRY(d_tri.alloc((size_t)n * TRI_F));
  SB_TRY(d_tri_out.alloc((size_t)n * TRI_F));
  SB_TRY(d_aux.alloc(n));
  for (int k = 0; k < 2; k++) {
    SB_TRY(d_order[k].alloc(n));
    SB_TRY(d_keys[k].alloc(n));
  }
  SB_TRY(d_head.alloc(n));
  SB_TRY(d_seg_of.alloc(n));
  SB_TRY(d_split_ids.alloc(n));
  SB_TRY(d_off_b.alloc(n));
  SB_TRY(d_off_e.alloc(n));
  SB_TRY(d_cost.alloc(n));
  SB_TRY(d_seg_min.alloc(n));
  SB_TRY(d_fwd.alloc(n));
  SB_TRY(d_rev.alloc(n));
  SB_TRY(d_pre.alloc(n));
  SB_TRY(d_suf.alloc(n));
  SB_TRY(d_node_pre.alloc(n));
  SB_TRY(d_fwd_key.alloc(n));
  SB_TRY(d_rev_key.alloc(n));
  SB_TRY(d_segs.alloc(n));
  SB_TRY(d_best.alloc(n));
  // one temporary buffer large enough for every rocPRIM call below
  size_t tmp_bytes = 0, need = 0;
  SB_TRY(rocprim::radix_sort_pairs(nullptr, need, d_keys[0].p, d_keys[1].p, d_order[0].p, d_order[1].p, (size_t)n, 0, 56,
                                   nullptr));
  tmp_bytes = need;
  SB_TRY(rocprim::inclusive_scan_by_key(nullptr, need, d_fwd_key.p, d_fwd.p, d_pre.p, (size_t)n, BoxOp(),
                                        rocprim::equal_to<uint32_t>(), nullptr));
  tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
  SB_TRY(rocprim::inclusive_scan(nullptr, need, d_head.p, d_seg_of.p, (size_t)n, rocprim::plus<int>(), nullptr));
  tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
  SB_TRY(rocprim::segmented_reduce(nullptr, need, d_cost.p, d_seg_min.p, (unsigned)n, d_off_b.p, d_off_e.p, U64Min(),
                                   ~0ull, nullptr));
  tmp_bytes = need > tmp_bytes ? need : tmp_bytes;
  SB_TRY(d_tmp.alloc(tmp_bytes));

  SB_TRY(hipMemcpy(d_tri.p, tri, (size_t)n * TRI_F * sizeof(float), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  SB_TRY(hipEventCreate(&e0));
  SB_TRY(hipEventCreate(&e1));
  SB_TRY(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL(k_aux, dim3(grid_n), dim3(TPB), 0, nullptr, d_tri.p, n, d_aux.p, d_order[0].p);
  int cur = 0; // d_order[cur] is the current array order

  // ---- level loop.  `nodes` is in creation order per level; ids in the reference's order come later.
  std::vector<HostNode> nodes;
  nodes.reserve((size_t)n / 2 + 16);
  std::vector<int> frontier_node; // node index of every frontier range, sorted by l
  std::vector<Seg> segs;
  {
    HostNode root;
    root.l = 0;
    root.r = n - 1;
    nodes.push_back(root);
    frontier_node.push_back(0);
  }
  std::vector<int> split_ids, off_b, off_e;
  std::vector<Best> best;
  std::vector<Box> node_box;
  std::vector<int> box_pos;
  std::vector<int> fresh; // nodes created at the previous level: their boxes are still to be read
  fresh.push_back(0);
  for (int level = 0; level < 4096; level++) {
    const int n_segs = (int)frontier_node.size();
    segs.resize((size_t)n_segs);
    split_ids.clear();
    off_b.clear();
    off_e.clear();
    for (int k = 0; k < n_segs; k++) {
      const HostNode& h = nodes[(size_t)frontier_node[(size_t)k]];
      Seg s;
      s.l = h.l;
      s.r = h.r;
      s.split = (h.r - h.l + 1) > leaf_n ? 1 : 0;
      s.axis = 0;
      segs[(size_t)k] = s;
      if (s.split) {
        split_ids.push_back(k);
        off_b.push_back(h.l);
        off_e.push_back(h.r); // candidates l .. r-1
      }
    }
    const int n_split = (int)split_ids.size();
    SB_TRY(hipMemcpyAsync(d_segs.p, segs.data(), (size_t)n_segs * sizeof(Seg), hipMemcpyHostToDevice, nullptr));
    SB_TRY(hipMemsetAsync(d_head.p, 0, (size_t)n * sizeof(int), nullptr));
    hipLaunchKernelGGL(k_heads, dim3((n_segs + TPB - 1) / TPB), dim3(TPB), 0, nullptr, d_segs.p, n_segs, d_head.p);
    need = tmp_bytes;
    SB_TRY(rocprim::inclusive_scan(d_tmp.p, need, d_head.p, d_seg_of.p, (size_t)n, rocprim::plus<int>(), nullptr));
    // boxes of the nodes created at the previous level (new_node's loop over [l, r] in the current order)
    hipLaunchKernelGGL(k_gather_boxes, dim3(grid_n), dim3(TPB), 0, nullptr, d_order[cur].p, d_aux.p, d_seg_of.p, d_segs.p, n,
                       d_fwd.p, d_fwd_key.p, d_rev.p, d_rev_key.p);
    need = tmp_bytes;
    SB_TRY(rocprim::inclusive_scan_by_key(d_tmp.p, need, d_fwd_key.p, d_fwd.p, d_node_pre.p, (size_t)n, BoxOp(),
                                          rocprim::equal_to<uint32_t>(), nullptr));
    if (!fresh.empty()) {
      const int m = (int)fresh.size();
      node_box.resize((size_t)m);
      box_pos.resize((size_t)m);
      for (int k = 0; k < m; k++) box_pos[(size_t)k] = nodes[(size_t)fresh[(size_t)k]].r; // prefix at r = the whole range
      SB_TRY(hipMemcpyAsync(d_off_b.p, box_pos.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, nullptr));
      hipLaunchKernelGGL(k_pick_boxes, dim3((m + TPB - 1) / TPB), dim3(TPB), 0, nullptr, d_node_pre.p, d_off_b.p, m, d_pre.p);
      SB_TRY(hipMemcpy(node_box.data(), d_pre.p, (size_t)m * sizeof(Box), hipMemcpyDeviceToHost));
      const float big = (float)1145141919;
      for (size_t k = 0; k < fresh.size(); k++) {
        HostNode& h = nodes[(size_t)fresh[k]];
        for (int c = 0; c < 3; c++) {
          h.AA[c] = gmin(big, node_box[k].lo[c]);
          h.BB[c] = gmax(-big, node_box[k].hi[c]);
        }
      }
    }
    if (n_split == 0) break;
    SB_TRY(hipMemcpyAsync(d_split_ids.p, split_ids.data(), (size_t)n_split * sizeof(int), hipMemcpyHostToDevice, nullptr));
    SB_TRY(hipMemcpyAsync(d_off_b.p, off_b.data(), (size_t)n_split * sizeof(int), hipMemcpyHostToDevice, nullptr));
    SB_TRY(hipMemcpyAsync(d_off_e.p, off_e.data(), (size_t)n_split * sizeof(int), hipMemcpyHostToDevice, nullptr));
    best.resize((size_t)n_split);
    if (median) {
      // buildBVH: sort along EVERY axis whose extent is >= the other two (three independent ifs: on ties the
      // last matching axis has the last word), split in the middle
      std::vector<Seg> pass(segs);
      for (int axis = 0; axis < 3; axis++) {
        bool any = false;
        for (int k = 0; k < n_segs; k++) {
          pass[(size_t)k].split = 0;
          if (!segs[(size_t)k].split) continue;
          const HostNode& h = nodes[(size_t)frontier_node[(size_t)k]];
          const float len[3] = {h.BB[0] - h.AA[0], h.BB[1] - h.AA[1], h.BB[2] - h.AA[2]};
          const int b = (axis + 1) % 3, c = (axis + 2) % 3;
          if (len[axis] >= len[b] && len[axis] >= len[c]) {
            pass[(size_t)k].split = 1;
            any = true;
          }
        }
        if (!any) continue;
        SB_TRY(hipMemcpy(d_segs.p, pass.data(), (size_t)n_segs * sizeof(Seg), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_keys, dim3(grid_n), dim3(TPB), 0, nullptr, d_segs.p, d_seg_of.p, d_order[cur].p, d_aux.p, n, axis,
                           d_keys[0].p);
        need = tmp_bytes;
        SB_TRY(rocprim::radix_sort_pairs(d_tmp.p, need, d_keys[0].p, d_keys[1].p, d_order[cur].p, d_order[cur ^ 1].p, (size_t)n,
                                         0, 56, nullptr));
        cur ^= 1;
      }
      for (int k = 0; k < n_split; k++) {
        const Seg& sg = segs[(size_t)split_ids[(size_t)k]];
        best[(size_t)k].cost = 0.0f;
        best[(size_t)k].axis = 0;
        best[(size_t)k].split = (sg.l + sg.r) / 2;
      }
    } else {
    for (int axis = 0; axis < 4; axis++) { // 0..2: the three sweeps; 3: the final sort on each range's best axis
      hipLaunchKernelGGL(k_keys, dim3(grid_n), dim3(TPB), 0, nullptr, d_segs.p, d_seg_of.p, d_order[cur].p, d_aux.p, n,
                         axis < 3 ? axis : -1, d_keys[0].p);
      need = tmp_bytes;
      SB_TRY(rocprim::radix_sort_pairs(d_tmp.p, need, d_keys[0].p, d_keys[1].p, d_order[cur].p, d_order[cur ^ 1].p, (size_t)n,
                                       0, 56, nullptr));
      cur ^= 1;
      if (axis == 3) break;
      hipLaunchKernelGGL(k_gather_boxes, dim3(grid_n), dim3(TPB), 0, nullptr, d_order[cur].p, d_aux.p, d_seg_of.p, d_segs.p,
                         n, d_fwd.p, d_fwd_key.p, d_rev.p, d_rev_key.p);
      need = tmp_bytes;
      SB_TRY(rocprim::inclusive_scan_by_key(d_tmp.p, need, d_fwd_key.p, d_fwd.p, d_pre.p, (size_t)n, BoxOp(),
                                            rocprim::equal_to<uint32_t>(), nullptr));
      need = tmp_bytes;
      SB_TRY(rocprim::inclusive_scan_by_key(d_tmp.p, need, d_rev_key.p, d_rev.p, d_suf.p, (size_t)n, BoxOp(),
                                            rocprim::equal_to<uint32_t>(), nullptr));
      hipLaunchKernelGGL(k_cost, dim3(grid_n), dim3(TPB), 0, nullptr, d_segs.p, d_seg_of.p, d_pre.p, d_suf.p, n, d_cost.p);
      need = tmp_bytes;
      SB_TRY(rocprim::segmented_reduce(d_tmp.p, need, d_cost.p, d_seg_min.p, (unsigned)n_split, d_off_b.p, d_off_e.p, U64Min(),
                                       ~0ull, nullptr));
      hipLaunchKernelGGL(k_best, dim3((n_split + TPB - 1) / TPB), dim3(TPB), 0, nullptr, d_seg_min.p, d_split_ids.p, d_segs.p,
                         n_split, axis, d_best.p);
      if (axis == 2)
        hipLaunchKernelGGL(k_set_axis, dim3((n_split + TPB - 1) / TPB), dim3(TPB), 0, nullptr, d_split_ids.p, d_best.p, n_split,
                           d_segs.p);
    }
    SB_TRY(hipMemcpy(best.data(), d_best.p, (size_t)n_split * sizeof(Best), hipMemcpyDeviceToHost));
    }
    // children, in range order
    std::vector<int> next;
    next.reserve((size_t)n_segs + (size_t)n_split);
    fresh.clear();
    int ks = 0;
    for (int k = 0; k < n_segs; k++) {
      const int id = frontier_node[(size_t)k];
      if (!segs[(size_t)k].split) {
        next.push_back(id);
        continue;
      }
      const Best b = best[(size_t)ks++];
      HostNode lc, rc;
      lc.l = nodes[(size_t)id].l;
      lc.r = b.split;
      rc.l = b.split + 1;
      rc.r = nodes[(size_t)id].r;
      nodes[(size_t)id].left = (int)nodes.size();
      nodes.push_back(lc);
      nodes[(size_t)id].right = (int)nodes.size();
      nodes.push_back(rc);
      fresh.push_back(nodes[(size_t)id].left);
      fresh.push_back(nodes[(size_t)id].right);
      next.push_back(nodes[(size_t)id].left);
      next.push_back(nodes[(size_t)id].right);
    }
    frontier_node.swap(next);
  }
  hipLaunchKernelGGL(k_gather_tris, dim3((unsigned)(((size_t)n * 9 + TPB - 1) / TPB)), dim3(TPB), 0, nullptr,
                     reinterpret_cast<const float4*>(d_tri.p), d_order[cur].p, n, reinterpret_cast<float4*>(d_tri_out.p));
  SB_TRY(hipEventRecord(e1, nullptr));
  SB_TRY(hipEventSynchronize(e1));
  SB_TRY(hipGetLastError());
  float ms = 0.0f;
  SB_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);

  // ---- ids in the reference's creation order (node, its whole left subtree, its right subtree); node 0 = testNode
  const int total = (int)nodes.size() + 1;
  if (total > nodes_capacity) {
    char buf[128];
    snprintf(buf, sizeof buf, "nodes_capacity %d too small: the tree has %d nodes", nodes_capacity, total);
    return ezrt_fail_msg(EZRT_ERR_INVALID, buf);
  }
  std::vector<int> id_of(nodes.size(), 0), stack;
  int next_id = 1;
  stack.push_back(0);
  while (!stack.empty()) {
    const int v = stack.back();
    stack.pop_back();
    id_of[(size_t)v] = next_id++;
    if (nodes[(size_t)v].left) {
      stack.push_back(nodes[(size_t)v].right);
      stack.push_back(nodes[(size_t)v].left);
    }
  }
  const float sentinel[EZRT_NODE_FLOATS] = {255.0f, 128.0f, 0.0f, 30.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f};
  for (int k = 0; k < EZRT_NODE_FLOATS; k++) nodes_out[k] = sentinel[k];
  for (size_t v = 0; v < nodes.size(); v++) {
    const HostNode& h = nodes[v];
    float* o = nodes_out + (size_t)id_of[v] * EZRT_NODE_FLOATS;
    const bool leaf = h.left == 0;
    o[0] = leaf ? 0.0f : (float)id_of[(size_t)h.left];
    o[1] = leaf ? 0.0f : (float)id_of[(size_t)h.right];
    o[2] = 0.0f;
    o[3] = leaf ? (float)(h.r - h.l + 1) : 0.0f;
    o[4] = leaf ? (float)h.l : 0.0f;
    o[5] = 0.0f;
    for (int c = 0; c < 3; c++) {
      o[6 + c] = h.AA[c];
      o[9 + c] = h.BB[c];
    }
  }
  SB_TRY(hipMemcpy(tri_out, d_tri_out.p, (size_t)n * TRI_F * sizeof(float), hipMemcpyDeviceToHost));
  *n_nodes = total;
  if (build_ms) *build_ms = ms;
  return 0;
}
} // namespace

extern "C" int ezrt_build_sah(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out,
                              int nodes_capacity, int* n_nodes, float* build_ms) {
  return build_level_sync(false, tri, n_tri, leaf_n, tri_out, nodes_out, nodes_capacity, n_nodes, build_ms);
}
extern "C" int ezrt_build_median(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out,
                                 int nodes_capacity, int* n_nodes, float* build_ms) {
  return build_level_sync(true, tri, n_tri, leaf_n, tri_out, nodes_out, nodes_capacity, n_nodes, build_ms);
}

extern "C" int ezrt_build_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError(); // (no device is an answer, not a sticky error)
    return 0;
  }
  return n > 0 ? n : 0;
}
