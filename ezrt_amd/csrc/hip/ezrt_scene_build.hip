// ezrt_scene_build.hip -- ezrt_scene_create / ezrt_scene_set_env: the device layout of a scene.  Host code and uploads only:
// the binary records of the caller's tree, the library's own binned-SAH tree over the reference's LEAVES (retree_leaves), its 4-wide
// collapse, the tables that order exact ties, the per-triangle pruning bounds (ezrt_traceq4.h "Distance pruning"), the env planes.
#include "ezrt_internal.h"

namespace {

// Host-side loops of ezrt_scene_create over triangles / leaves (round 4: C5's create took 0.3 s on one thread).  Chunks of
// [0, n) on up to 16 threads; every use writes disjoint elements and reduces with order-independent operations (max, integer
// sums), so the result does not depend on the thread count.
template <class F>
void parallel_for(int n, int grain, F f) {
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = getenv("EZRT_HOST_THREADS")) hw = (unsigned)std::max(1, atoi(e));
  int nt = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
  nt = std::min(nt, std::max(1, n / std::max(1, grain)));
  if (nt <= 1) {
    f(0, n, 0);
    return;
  }
  // Nothing may leave through the C ABI as an exception: a worker's exception (std::bad_alloc in a lambda's vector)
  // is carried to the caller's thread, a thread that cannot be created (std::system_error under a thread cap) has its range
  // run inline; every thread that did start is joined before anything is rethrown -- ezrt_scene_create turns it into an error code.
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nt);
  th.reserve((size_t)nt);
  for (int k = 0; k < nt; k++) {
    const int lo = (int)((long long)n * k / nt), hi = (int)((long long)n * (k + 1) / nt);
    auto body = [=, &f, &err] {
      try {
        f(lo, hi, k);
      } catch (...) {
        err[(size_t)k] = std::current_exception();
      }
    };
    try {
      th.emplace_back(body);
    } catch (const std::system_error&) {
      body();
    }
  }
  for (auto& t : th) t.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}
constexpr int PAR_MAX = 16; // threads of parallel_for at most (per-thread partial results are arrays of this size)

struct HostNode {
  int left, right, n, index;
  float AA[3], BB[3];
};
HostNode decode_node(const float* nodes, int i) {
  const float* p = nodes + (size_t)i * EZRT_NODE_FLOATS;
  HostNode h;
  h.left = (int)p[0]; // ivec3(texelFetch) truncation, P5/fsh:143-148
  h.right = (int)p[1];
  h.n = (int)p[3];
  h.index = (int)p[4];
  for (int k = 0; k < 3; k++) {
    h.AA[k] = p[6 + k];
    h.BB[k] = p[9 + k];
  }
  return h;
}

// ---- retree_leaves: OUR tree over the REFERENCE'S leaves (round 3).
// For a tame ray and nested boxes the fp32 slab test is monotone (ezrt_traceq4.h), so the reference's hitBVH reaches a leaf
// iff the slab test of the leaf's OWN box says hit: every ancestor's box contains it and is hit a fortiori.  The set of
// leaves a ray visits -- and with it the set of triangles tested, the minimum of t, the exact ties -- therefore does not
// depend on the inner nodes at all: ANY tree whose inner boxes are unions of the reference's leaf boxes visits exactly the
// same leaves.  The reference's inner nodes are poor where its builder hits its `INF = 114514` cost cap (P3/main.cpp:492,
// 538: the node silently becomes a median-x split; 180 nodes of the 10^6-triangle scene, all at the top), so the device
// layout builds its own: a top-down binned SAH (32 bins per axis, cost = area x triangle count) over the reference's leaf
// boxes, then the same 4-wide collapse.  The leaves -- boxes, triangle ranges, order -- are the reference's, untouched; the
// binary records of the in-order kernel (redo launches, instrumented runs, counters P/I/T/M) stay the reference's tree.
struct LeafPrim {
  float c[3];
  int node, w;
  float AA[3], BB[3]; // the leaf's box (a copy: the binning loop streams these instead of chasing `node` into the reference's array)
};
inline void retree_union(std::vector<HostNode>& out, int id) { // exact unions: every box is nested in its parent's by construction
  HostNode& h = out[(size_t)id];
  for (int k = 0; k < 3; k++) {
    h.AA[k] = std::min(out[(size_t)h.left].AA[k], out[(size_t)h.right].AA[k]);
    h.BB[k] = std::max(out[(size_t)h.left].BB[k], out[(size_t)h.right].BB[k]);
  }
}
struct RetreeJob { // a subrange whose subtree is built by another thread and spliced in afterwards
  int slot, begin, end, depth;
};
// jobs (or NULL): ranges of at most `grain` leaves below the first two levels are not built here but listed, their root an empty slot
int retree_build(std::vector<LeafPrim>& pr, int begin, int end, const std::vector<HostNode>& ref, std::vector<HostNode>& out, int depth = 0,
                 std::vector<RetreeJob>* jobs = nullptr, int grain = 0) {
  const int id = (int)out.size();
  out.push_back(HostNode());
  if (jobs && depth >= 2 && end - begin <= grain && end - begin > 1) {
    jobs->push_back(RetreeJob{id, begin, end, depth});
    return id;
  }
  if (end - begin == 1) {
    out[(size_t)id] = ref[(size_t)pr[(size_t)begin].node];
    out[(size_t)id].left = out[(size_t)id].right = 0;
    return id;
  }
  float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = begin; i < end; i++)
    for (int a = 0; a < 3; a++) {
      clo[a] = std::min(clo[a], pr[(size_t)i].c[a]);
      chi[a] = std::max(chi[a], pr[(size_t)i].c[a]);
    }
  constexpr int NBMAX = 64;
  // (16, 32 and 64 bins, and an exact sweep over sorted centroids for small or for all ranges, were within +-2 % of each
  // other on C2 / C3 / C5: 32 bins)
  static const int NB = [] { const char* e = getenv("EZRT_RETREE_BINS"); int v = e ? atoi(e) : 32; return v < 2 ? 2 : (v > NBMAX ? NBMAX : v); }();
  double best = 1e300;
  int best_axis = -1, best_split = -1;
  for (int a = 0; a < 3; a++) {
    const float ext = chi[a] - clo[a];
    if (!(ext > 0.0f)) continue;
    float lo[NBMAX][3], hi[NBMAX][3];
    long long cnt[NBMAX];
    for (int b = 0; b < NB; b++) {
      cnt[b] = 0;
      for (int k = 0; k < 3; k++) lo[b][k] = 3.0e38f, hi[b][k] = -3.0e38f;
    }
    const float scale = (float)NB / ext;
    for (int i = begin; i < end; i++) {
      int b = (int)((pr[(size_t)i].c[a] - clo[a]) * scale);
      b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
      const LeafPrim& h = pr[(size_t)i];
      cnt[b] += h.w;
      for (int k = 0; k < 3; k++) {
        lo[b][k] = std::min(lo[b][k], h.AA[k]);
        hi[b][k] = std::max(hi[b][k], h.BB[k]);
      }
    }
    // sweep: suffix boxes, then prefix
    double ra[NBMAX];
    long long rc[NBMAX];
    float slo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, shi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    long long c = 0;
    auto area = [](const float* l, const float* h) {
      const double ex = (double)h[0] - l[0], ey = (double)h[1] - l[1], ez = (double)h[2] - l[2];
      return (ex < 0 || ey < 0 || ez < 0) ? 0.0 : 2.0 * (ex * ey + ey * ez + ez * ex);
    };
    for (int b = NB - 1; b >= 1; b--) {
      for (int k = 0; k < 3; k++) slo[k] = std::min(slo[k], lo[b][k]), shi[k] = std::max(shi[k], hi[b][k]);
      c += cnt[b];
      ra[b] = area(slo, shi);
      rc[b] = c;
    }
    float plo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, phi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    long long lc = 0;
    for (int b = 0; b < NB - 1; b++) { // split after bin b
      for (int k = 0; k < 3; k++) plo[k] = std::min(plo[k], lo[b][k]), phi[k] = std::max(phi[k], hi[b][k]);
      lc += cnt[b];
      if (lc == 0 || rc[b + 1] == 0) continue;
      const double cost = area(plo, phi) * (double)lc + ra[b + 1] * (double)rc[b + 1];
      if (cost < best) {
        best = cost;
        best_axis = a;
        best_split = b;
      }
    }
  }
  int mid;
  if (best_axis >= 0 && depth < 40) { // (below 40 levels of SAH splits: medians, so that the depth stays bounded)
    const float ext = chi[best_axis] - clo[best_axis], scale = (float)NB / ext, lo0 = clo[best_axis];
    const int a = best_axis, sp = best_split;
    auto it = std::partition(pr.begin() + begin, pr.begin() + end, [&](const LeafPrim& q) {
      int b = (int)((q.c[a] - lo0) * scale);
      b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
      return b <= sp;
    });
    mid = (int)(it - pr.begin());
  } else {
    mid = begin; // (all centroids equal, or no bin boundary separates them)
  }
  if (mid <= begin || mid >= end) { // object median along the widest centroid axis
    int a = 0;
    if (chi[1] - clo[1] > chi[a] - clo[a]) a = 1;
    if (chi[2] - clo[2] > chi[a] - clo[a]) a = 2;
    mid = (begin + end) / 2;
    std::nth_element(pr.begin() + begin, pr.begin() + mid, pr.begin() + end,
                     [a](const LeafPrim& x, const LeafPrim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.node < y.node); });
  }
  const int l = retree_build(pr, begin, mid, ref, out, depth + 1, jobs, grain), r = retree_build(pr, mid, end, ref, out, depth + 1, jobs, grain);
  HostNode& h = out[(size_t)id];
  h.left = l;
  h.right = r;
  h.n = 0;
  h.index = 0;
  if (!jobs) retree_union(out, id); // (with deferred subtrees below, the unions are taken once they are spliced in: retree_leaves)
  return id;
}
// tree[0] dummy, tree[1] root, children after parents; leaves are copies of the reference's reachable leaves
bool retree_leaves(const std::vector<HostNode>& ref, int n_nodes, std::vector<HostNode>& tree) {
  std::vector<LeafPrim> pr;
  std::vector<int> todo(1, 1);
  std::vector<char> seen((size_t)n_nodes, 0);
  while (!todo.empty()) { // reachable leaves (the arrays are a tree here: checked by the caller)
    const int i = todo.back();
    todo.pop_back();
    if (seen[(size_t)i]) continue;
    seen[(size_t)i] = 1;
    const HostNode& h = ref[(size_t)i];
    if (h.n > 0) {
      LeafPrim q;
      for (int k = 0; k < 3; k++) {
        q.c[k] = 0.5f * h.AA[k] + 0.5f * h.BB[k];
        q.AA[k] = h.AA[k];
        q.BB[k] = h.BB[k];
      }
      q.node = i;
      q.w = h.n;
      pr.push_back(q);
    } else {
      todo.push_back(h.right);
      todo.push_back(h.left);
    }
  }
  if (pr.size() < 2) return false;
  for (const LeafPrim& q : pr)
    for (int k = 0; k < 3; k++)
      if (!(q.c[k] > -3.0e38f && q.c[k] < 3.0e38f)) return false; // (non-finite boxes: keep the reference's tree)
  tree.clear();
  tree.reserve(2 * pr.size() + 1);
  tree.push_back(HostNode());
  // The top of the tree is built here; subtrees of at most 1/32 of the leaves are built by worker threads into vectors of their
  // own (disjoint ranges of `pr`) and spliced in behind it -- any numbering with children after their parents will do, and
  // the tree itself does not depend on the thread count (the same splits, the same unions).
  std::vector<RetreeJob> jobs;
  const int grain = pr.size() >= 65536 ? (int)(pr.size() / 32) : 0;
  retree_build(pr, 0, (int)pr.size(), ref, tree, 0, grain ? &jobs : nullptr, grain);
  if (grain) {
    std::vector<std::vector<HostNode>> local(jobs.size());
    parallel_for((int)jobs.size(), 1, [&](int lo, int hi, int) {
      for (int j = lo; j < hi; j++) {
        local[(size_t)j].reserve(2 * (size_t)(jobs[(size_t)j].end - jobs[(size_t)j].begin));
        retree_build(pr, jobs[(size_t)j].begin, jobs[(size_t)j].end, ref, local[(size_t)j], jobs[(size_t)j].depth);
      }
    });
    const int top_count = (int)tree.size(); // nodes made by this thread: ids [1, top_count), the job slots among them
    std::vector<char> is_job((size_t)top_count, 0);
    for (const RetreeJob& j : jobs) is_job[(size_t)j.slot] = 1;
    for (size_t j = 0; j < jobs.size(); j++) { // local index 0 = the job's slot, k > 0 -> base + k - 1
      const std::vector<HostNode>& L = local[j];
      const int base = (int)tree.size(), slot = jobs[j].slot;
      auto map = [&](int k) { return k == 0 ? slot : base + k - 1; };
      for (size_t k = 0; k < L.size(); k++) {
        HostNode h = L[k];
        if (h.n <= 0) {
          h.left = map(h.left);
          h.right = map(h.right);
        }
        if (k == 0) tree[(size_t)slot] = h;
        else tree.push_back(h);
      }
    }
    // boxes of the top nodes: children carry larger ids than their parents, so one backward sweep over the inner nodes this
    // thread made (their unions were postponed: the job slots had no box yet)
    for (int i = top_count - 1; i >= 1; i--)
      if (!is_job[(size_t)i] && tree[(size_t)i].n <= 0) retree_union(tree, i);
  }
  return true;
}

} // namespace

extern "C" {

static int scene_create_impl(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out);
int ezrt_scene_create(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out) {
  if (!out) return fail(EZRT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  return ezi::guarded("ezrt_scene_create", [&]() -> int { return scene_create_impl(tri, n_tri, nodes, n_nodes, out); });
}
static int scene_create_impl(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out) {
  if (!tri || !nodes || n_tri <= 0 || n_nodes <= 0) return fail(EZRT_ERR_INVALID, "empty scene arrays");
  if (n_tri >= (1 << 24) || n_nodes >= (1 << 24))
    return fail(EZRT_ERR_UNSUPPORTED, "counts >= 2^24 are not exact in the float encoding");
  if (n_nodes < 2) return fail(EZRT_ERR_INVALID, "need at least the dummy node 0 and the root node 1");
  // EZRT_CREATE_TIMING=1: wall time of the host-side phases below, to stderr
  static const bool timing = getenv("EZRT_CREATE_TIMING") && atoi(getenv("EZRT_CREATE_TIMING")) != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[ezrt] scene_create %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };

  // ---- validate + measure the tree (pre-order ids: child > parent)
  std::vector<int> depth((size_t)n_nodes, 0), inner_id((size_t)n_nodes, -1);
  int64_t leaves = 0, maxleaf = 0;
  int n_inner = 0;
  for (int i = 1; i < n_nodes; i++) {
    HostNode h = decode_node(nodes, i);
    if (h.n > 0) {
      if (h.index < 0 || (int64_t)h.index + h.n > n_tri)
        return fail(EZRT_ERR_INVALID, "leaf %d: triangle range outside the triangle array", i);
      leaves++;
      if (h.n > maxleaf) maxleaf = h.n;
    } else {
      if (h.left <= i || h.right <= i || h.left >= n_nodes || h.right >= n_nodes)
        return fail(EZRT_ERR_INVALID, "inner node %d: children must satisfy parent < child < nNodes", i);
      inner_id[(size_t)i] = n_inner++;
    }
  }
  int maxd = 1;
  depth[1] = 1;
  for (int i = 1; i < n_nodes; i++) {
    if (depth[(size_t)i] == 0) continue;
    HostNode h = decode_node(nodes, i);
    if (h.n <= 0) {
      // caller arrays may reference a node from several parents (a DAG): the LDS stack must fit the
      // DEEPEST path, so keep the maximum (ids are topologically ordered: one pass is exact)
      depth[(size_t)h.left] = std::max(depth[(size_t)h.left], depth[(size_t)i] + 1);
      depth[(size_t)h.right] = std::max(depth[(size_t)h.right], depth[(size_t)i] + 1);
    }
    if (depth[(size_t)i] > maxd) maxd = depth[(size_t)i];
  }
  if (maxd + 1 > 256) return fail(EZRT_ERR_UNSUPPORTED, "tree deeper than the reference's 256-entry stack");
  if (maxleaf > 128) return fail(EZRT_ERR_UNSUPPORTED, "leaf with %lld triangles: this build packs leaf size in 7 bits (<= 128)", (long long)maxleaf);
  // per-lane traversal stack in LDS: depth rows of 256 ints + the lane table must fit the 64 KiB a workgroup gets
  // without an opt-in (every kernel that walks the tree is launched with that much dynamic LDS at most)
  if (((size_t)maxd + 1) * BLOCK * sizeof(int) > 64 * 1024)
    return fail(EZRT_ERR_UNSUPPORTED, "tree depth %d: the LDS traversal stack of this build holds depth <= 63 (the reference's is 256 entries; "
                "its builders reach depth ~30 on 10^6 triangles)", maxd);

  // ---- device layout.  Inner records are numbered breadth-first from the root (ids are internal
  // to the device layout) so that records [0, K) are the top levels of the tree: traceq_kernel stages
  // that prefix in LDS.  Unreachable inner nodes (never visited) go last.
  {
    std::vector<int> order;
    order.reserve((size_t)n_inner);
    std::vector<char> seen((size_t)n_nodes, 0);
    if (inner_id[1] >= 0) {
      order.push_back(1);
      seen[1] = 1;
    }
    for (size_t q = 0; q < order.size(); q++) {
      HostNode h = decode_node(nodes, order[q]);
      const int kids[2] = {h.left, h.right};
      for (int c : kids)
        if (inner_id[(size_t)c] >= 0 && !seen[(size_t)c]) {
          seen[(size_t)c] = 1;
          order.push_back(c);
        }
    }
    for (int i = 1; i < n_nodes; i++)
      if (inner_id[(size_t)i] >= 0 && !seen[(size_t)i]) order.push_back(i);
    for (size_t q = 0; q < order.size(); q++) inner_id[(size_t)order[q]] = (int)q;
  }
  lap("validate + numbering");
  auto ref_of = [&](int node) -> uint32_t {
    HostNode h = decode_node(nodes, node);
    if (h.n > 0) return LEAF_BIT | ((uint32_t)(h.n - 1) << 24) | (uint32_t)h.index;
    return (uint32_t)inner_id[(size_t)node];
  };
  std::vector<float4> inner((size_t)(n_inner > 0 ? n_inner : 1) * 4);
  for (int i = 1; i < n_nodes; i++) {
    if (inner_id[(size_t)i] < 0) continue;
    HostNode h = decode_node(nodes, i);
    HostNode l = decode_node(nodes, h.left), r = decode_node(nodes, h.right);
    float4* q = &inner[(size_t)inner_id[(size_t)i] * 4];
    q[0] = make_float4(l.AA[0], l.AA[1], l.AA[2], l.BB[0]);
    q[1] = make_float4(l.BB[1], l.BB[2], r.AA[0], r.AA[1]);
    q[2] = make_float4(r.AA[2], r.BB[0], r.BB[1], r.BB[2]);
    uint32_t lr = ref_of(h.left), rr = ref_of(h.right);
    float lf, rf;
    memcpy(&lf, &lr, 4);
    memcpy(&rf, &rr, 4);
    q[3] = make_float4(lf, rf, 0.0f, 0.0f);
  }
  lap("binary records");
  // ---- 4-wide collapse for traceq4_kernel (ezrt_traceq4.h).  Valid only when every box is nested in its
  // parent's box (true for the reference builders; checked here because the arrays are the caller's).
  std::vector<float4> inner4;
  std::vector<std::array<int, 4>> rec_slot_nodes; // per record (in its final numbering): tree4's node of each slot, 0 = unused
  std::vector<HostNode> tree4;                    // the binary tree the records are a collapse of: the reference's, or retree_leaves'
  bool retreed = false;
  int n_inner4 = 0, stack_need4 = 1;
  {
    std::vector<HostNode> hn((size_t)n_nodes);
    for (int i = 1; i < n_nodes; i++) hn[(size_t)i] = decode_node(nodes, i);
    lap("  decode nodes");
    bool nested = inner_id[1] >= 0;
    // caller arrays may be a DAG (an inner node referenced by several parents: validation only asks parent < child).
    // The collapse below makes one record per (parent, inner child) visit and indexes records by node, so a shared
    // node would get two records, one of them never numbered (a write before the vector's buffer) and
    // chains of shared nodes would multiply records.  Such arrays keep the binary kernel.
    {
      std::vector<unsigned char> n_parents((size_t)n_nodes, 0);
      for (int i = 1; i < n_nodes && nested; i++) {
        if (inner_id[(size_t)i] < 0) continue;
        const int kids[2] = {hn[(size_t)i].left, hn[(size_t)i].right};
        for (int k : kids) // (leaves too: the tie tables hold ONE parent per node and the re-tree visits a leaf once -- ADVICE r3)
          if (++n_parents[(size_t)k] > 1) nested = false;
      }
    }
    for (int i = 2; i < n_nodes && nested; i++) { // (the root's own box is never tested)
      if (inner_id[(size_t)i] < 0) continue;
      const HostNode& c = hn[(size_t)i];
      const int kids[2] = {c.left, c.right};
      for (int k : kids)
        for (int ax = 0; ax < 3; ax++)
          if (!(hn[(size_t)k].AA[ax] >= c.AA[ax] && hn[(size_t)k].BB[ax] <= c.BB[ax])) nested = false; // (false on NaN)
    }
    // (two attempts at most: if the library's own tree over the leaves comes out so deep that the 4-wide kernel's stack rows
    // would not fit its LDS -- use_wide4 -- the records are rebuilt as a cut of the CALLER's inner nodes, which may fit: ADVICE r3)
    for (int attempt = 0; nested && attempt < 2; attempt++) {
      retreed = attempt == 0 && tuning_from_env().retree != 0 && retree_leaves(hn, n_nodes, tree4);
      if (!retreed) tree4 = hn; // (node ids = the caller's)
      lap("  retree_leaves");
      auto is_inner = [&](int i) { return tree4[(size_t)i].n <= 0; };
      auto area = [&](int i) { // schedule heuristic only
        const HostNode& h = tree4[(size_t)i];
        float ex = h.BB[0] - h.AA[0], ey = h.BB[1] - h.AA[1], ez = h.BB[2] - h.AA[2];
        float a = ex * ey + ey * ez + ez * ex;
        return a == a ? a : 0.0f;
      };
      auto leaf_pair = [&](int i) { return is_inner(i) && !is_inner(tree4[(size_t)i].left) && !is_inner(tree4[(size_t)i].right); };
      struct Rec {
        int node, m, slot[4];
      };
      // a record per reachable "cut root"; slots = a cut of <= 4 descendants: start from the two children and keep
      // splitting an inner slot (first a pair of leaves -- it would otherwise become a half-empty record of
      // its own -- else the one with the largest box) while there is room
      std::vector<Rec> recs;
      std::vector<int> rec_of(tree4.size(), -1);
      std::vector<int> todo(1, 1);
      while (!todo.empty()) {
        const int x = todo.back();
        todo.pop_back();
        Rec r;
        r.node = x;
        r.m = 2;
        r.slot[0] = tree4[(size_t)x].left;
        r.slot[1] = tree4[(size_t)x].right;
        while (r.m < 4) {
          int pick = -1;
          bool pick_pair = false;
          for (int k = 0; k < r.m; k++) {
            const int g = r.slot[k];
            if (!is_inner(g)) continue;
            const bool pr = leaf_pair(g);
            if (pick < 0 || (pr && !pick_pair) || (pr == pick_pair && area(g) > area(r.slot[pick]))) {
              pick = k;
              pick_pair = pr;
            }
          }
          if (pick < 0) break;
          const int g = r.slot[pick];
          r.slot[pick] = tree4[(size_t)g].left;
          r.slot[r.m++] = tree4[(size_t)g].right;
        }
        rec_of[(size_t)x] = (int)recs.size();
        recs.push_back(r);
        for (int k = 0; k < r.m; k++)
          if (is_inner(r.slot[k])) todo.push_back(r.slot[k]);
      }
      // stack rows a subtree can need: slots are visited in ascending order (the lowest hit slot next, the others
      // pushed highest-first), so slot j is entered with at most m-1-j entries pending: need = max_j(m-1-j + need_j);
      // minimised by ascending need.  Children were created after their parents: walk the records backwards.
      std::vector<int> need(recs.size(), 0);
      for (size_t q = recs.size(); q-- > 0;) {
        Rec& r = recs[q];
        int nd[4];
        for (int k = 0; k < r.m; k++) nd[k] = is_inner(r.slot[k]) ? need[(size_t)rec_of[(size_t)r.slot[k]]] : 0;
        for (int i = 1; i < r.m; i++) // insertion sort by need, stable
          for (int j = i; j > 0 && nd[j - 1] > nd[j]; j--) {
            std::swap(nd[j - 1], nd[j]);
            std::swap(r.slot[j - 1], r.slot[j]);
          }
        int w = 0;
        for (int j = 0; j < r.m; j++) w = std::max(w, r.m - 1 - j + nd[j]);
        need[q] = w;
      }
      lap("  cuts + stack need");
      stack_need4 = std::max(1, need[0]);
      // breadth-first numbering: the top of the tree is a prefix (staged in LDS)
      std::vector<int> order(1, 0), number(recs.size(), -1);
      number[0] = 0;
      for (size_t q = 0; q < order.size(); q++) {
        const Rec& r = recs[(size_t)order[q]];
        for (int k = 0; k < r.m; k++)
          if (is_inner(r.slot[k])) {
            const int c = rec_of[(size_t)r.slot[k]];
            number[(size_t)c] = (int)order.size();
            order.push_back(c);
          }
      }
      n_inner4 = (int)recs.size();
      inner4.assign((size_t)n_inner4 * N4_FLOAT4, make_float4(0, 0, 0, 0));
      rec_slot_nodes.assign((size_t)n_inner4, std::array<int, 4>{0, 0, 0, 0});
      const float qnan = __builtin_nanf("");
      for (size_t q = 0; q < recs.size(); q++) {
        const Rec& r = recs[q];
        float v[7][4];
        for (int k = 0; k < 4; k++) {
          uint32_t rf = REF_EMPTY;
          for (int c = 0; c < 6; c++) v[c][k] = qnan; // unused slot: never hit (see ezrt_traceq4.h)
          if (k < r.m) {
            const HostNode& g = tree4[(size_t)r.slot[k]];
            for (int c = 0; c < 3; c++) {
              v[c][k] = g.AA[c];
              v[3 + c][k] = g.BB[c];
            }
            rf = is_inner(r.slot[k]) ? (uint32_t)number[(size_t)rec_of[(size_t)r.slot[k]]]
                                     : (LEAF_BIT | ((uint32_t)(g.n - 1) << 24) | (uint32_t)g.index);
          }
          memcpy(&v[6][k], &rf, 4);
        }
        if (number[q] < 0) return fail(EZRT_ERR_INVALID, "internal: 4-wide record %zu of node %d was never numbered", q, r.node);
        for (int k = 0; k < r.m; k++) rec_slot_nodes[(size_t)number[q]][k] = r.slot[k];
        float4* o = &inner4[(size_t)number[q] * N4_FLOAT4];
        for (int c = 0; c < 3; c++) { // rows: see EZRT_SLAB_SELECT in ezrt_traceq4.h
          o[N4_ROW_AA + c] = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
          o[N4_ROW_BB + c] = make_float4(v[3 + c][0], v[3 + c][1], v[3 + c][2], v[3 + c][3]);
        }
        o[N4_ROW_REF] = make_float4(v[6][0], v[6][1], v[6][2], v[6][3]);
      }
      if (!retreed || ((size_t)stack_need4 + 4) * BLOCK * sizeof(int) <= 60 * 1024) break; // (the bound of use_wide4)
    }
  }
  lap("re-tree + 4-wide collapse");
  // ---- tables of tie_precedes (ezrt_traceq4.h): only for arrays that are a tree with nested boxes (the 4-wide records exist)
  // and whose leaves do not share triangles
  std::vector<int32_t> tri_leaf_h;
  std::vector<int2> ref_up_h;
  if (n_inner4 > 0) {
    tri_leaf_h.assign((size_t)n_tri, -1);
    ref_up_h.assign((size_t)n_nodes, make_int2(0, 0));
    bool ok = true;
    for (int i = 1; i < n_nodes && ok; i++) {
      if (depth[(size_t)i] == 0) continue; // unreachable
      const HostNode h = decode_node(nodes, i);
      if (h.n > 0) {
        for (int k = h.index; k < h.index + h.n; k++) {
          if (tri_leaf_h[(size_t)k] >= 0) ok = false; // a triangle in two leaves: no unique leaf
          tri_leaf_h[(size_t)k] = i;
        }
      } else {
        const int kids[2] = {h.left, h.right};
        for (int c = 0; c < 2; c++)
          ref_up_h[(size_t)kids[c]] = make_int2((int)((uint32_t)i | ((uint32_t)depth[(size_t)kids[c]] << 24)),
                                               (int)((uint32_t)inner_id[(size_t)i] | (c ? 0x80000000u : 0u)));
      }
    }
    ref_up_h[1] = make_int2((int)(1u << 24), 0);
    if (!ok) {
      tri_leaf_h.clear();
      ref_up_h.clear();
    } else {
      for (int32_t& v : tri_leaf_h)
        if (v < 0) v = 1; // (triangles no leaf holds are never tested)
    }
  }
  lap("tie tables");
  std::vector<float4> geom((size_t)n_tri * 3);
  parallel_for(n_tri, 1 << 15, [&](int lo_i, int hi_i, int) {
  for (int i = lo_i; i < hi_i; i++) {
    const float* t = tri + (size_t)i * EZRT_TRI_FLOATS;
    // N = normalize(cross(p2 - p1, p3 - p1)), P5/fsh:172 -- same fp32 ops, contraction off
    float e1x = t[3] - t[0], e1y = t[4] - t[1], e1z = t[5] - t[2];
    float e2x = t[6] - t[0], e2y = t[7] - t[1], e2z = t[8] - t[2];
    float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
    float inv = 1.0f / __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
    geom[(size_t)i * 3 + 0] = make_float4(t[0], t[1], t[2], cx * inv);
    geom[(size_t)i * 3 + 1] = make_float4(t[3], t[4], t[5], cy * inv);
    geom[(size_t)i * 3 + 2] = make_float4(t[6], t[7], t[8], cz * inv);
  }
  });

  lap("geometry records");
  // ---- distance pruning (ezrt_traceq4.h "Distance pruning"): the per-triangle bound eta_T in double precision, A = 2 max
  // eta_T over the triangles below each slot (row 7 of the 4-wide records).  Leaf boxes must hold their triangles (true for
  // the reference builders; these are the caller's arrays).
  bool prunable = n_inner4 > 0;
  double prune_G = 0.0, prune_Z = 0.0, prune_M = 0.0, prune_A_med = 0.0;
  float prune_a = 0.0f;
  uint32_t root4_flag = 0u;
  int64_t prune_bad = 0, prune_flagged = 0;
  if (prunable) {
    for (int i = 1; i < n_nodes && prunable; i++) {
      const HostNode h = decode_node(nodes, i);
      if (h.n <= 0) continue;
      for (int k = h.index; k < h.index + h.n && prunable; k++) {
        const float* t = tri + (size_t)k * EZRT_TRI_FLOATS;
        for (int v = 0; v < 9; v++)
          if (!(t[v] >= h.AA[v % 3] && t[v] <= h.BB[v % 3])) prunable = false; // (false on NaN)
      }
    }
  }
  lap("  leaf boxes hold their triangles");
  if (prunable) {
    const double eps = 1.0 / 16777216.0, dinf = (double)__builtin_inff();
    std::vector<double> eta((size_t)n_tri, 0.0);
    double part_M[PAR_MAX] = {0}, part_G[PAR_MAX] = {0}, part_Z[PAR_MAX] = {0}; // per-thread maxima and counts (order-independent)
    int64_t part_bad[PAR_MAX] = {0};
    parallel_for(n_tri, 1 << 14, [&](int lo_i, int hi_i, int tid) {
    double prune_M = 0.0, prune_G = 0.0, prune_Z = 0.0; // (this thread's)
    int64_t prune_bad = 0;
    for (int i = lo_i; i < hi_i; i++) {
      const float* t = tri + (size_t)i * EZRT_TRI_FLOATS;
      double p[3][3], n[3] = {(double)geom[(size_t)i * 3].w, (double)geom[(size_t)i * 3 + 1].w, (double)geom[(size_t)i * 3 + 2].w}, m_t = 0.0;
      for (int v = 0; v < 3; v++)
        for (int c = 0; c < 3; c++) {
          p[v][c] = (double)t[v * 3 + c];
          m_t = __builtin_fmax(m_t, p[v][c] < 0 ? -p[v][c] : p[v][c]);
        }
      prune_M = __builtin_fmax(prune_M, m_t);
      const double nn = __builtin_sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(nn == nn) || nn > 1e300 || nn == 0.0) continue; // NaN / inf / zero normal: hit_triangle_t can never accept it (eta = 0)
      eta[(size_t)i] = dinf;                                  // until proven otherwise
      if (!(m_t < 1e30) || !(nn > 0.5 && nn < 2.0)) { // a stored normal that is not unit (underflow in the cross product): no bound
        prune_bad++;
        continue;
      }
      double u[3] = {n[0] / nn, n[1] / nn, n[2] / nn}, q[3][3], zeta = 0.0;
      for (int v = 0; v < 3; v++) {
        const double h = u[0] * (p[v][0] - p[0][0]) + u[1] * (p[v][1] - p[0][1]) + u[2] * (p[v][2] - p[0][2]);
        for (int c = 0; c < 3; c++) q[v][c] = p[v][c] - u[c] * h;
        zeta = __builtin_fmax(zeta, h < 0 ? -h : h);
      }
      double smin = 1.0, diam = 0.0, emin = dinf; // min sin(angle / 2), longest and shortest edge of the projected triangle
      for (int v = 0; v < 3; v++) {
        const double* o = q[v];
        const double* e = q[(v + 1) % 3];
        const double* f = q[(v + 2) % 3];
        const double a[3] = {e[0] - o[0], e[1] - o[1], e[2] - o[2]}, b[3] = {f[0] - o[0], f[1] - o[1], f[2] - o[2]};
        const double la = __builtin_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), lb = __builtin_sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        diam = __builtin_fmax(diam, la);
        emin = __builtin_fmin(emin, la);
        if (!(la > 0.0 && lb > 0.0)) {
          smin = 0.0;
          break;
        }
        double c = (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) / (la * lb);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        smin = __builtin_fmin(smin, __builtin_sqrt((1.0 - c) * 0.5));
      }
      if (!(smin >= 1e-4) || !(zeta <= 1e-3 * emin)) { // thinner than ~0.01 degrees, or bent off its stored plane: no bound
        prune_bad++;
        continue;
      }
      eta[(size_t)i] = zeta + 15.2 * eps * (diam + zeta) / smin + 19.8 * eps * m_t;
      prune_G = __builtin_fmax(prune_G, 1.0 / smin);
      prune_Z = __builtin_fmax(prune_Z, zeta);
    }
    part_M[tid] = prune_M;
    part_G[tid] = prune_G;
    part_Z[tid] = prune_Z;
    part_bad[tid] = prune_bad;
    });
    for (int k = 0; k < PAR_MAX; k++) {
      prune_M = __builtin_fmax(prune_M, part_M[k]);
      prune_G = __builtin_fmax(prune_G, part_G[k]);
      prune_Z = __builtin_fmax(prune_Z, part_Z[k]);
      prune_bad += part_bad[k];
    }
  lap("  eta loop");
    // ordinary triangles: eta_T <= cutoff.  cutoff = 2^-13 max|coordinate| when that already leaves a margin that is small
    // at the scale the geometry lives on (<= 2^-14 of the median triangle's largest coordinate); otherwise -- a ground
    // plane of kilometres under a metre-sized model -- the 1 - 2^-10 quantile of the bounds.  The others flag every node above them.
    double cutoff = prune_M / 8192.0;
    {
      std::vector<double> all, mts;
      double a_glob = 0.0;
      for (int i = 0; i < n_tri; i++) {
        const float* t = tri + (size_t)i * EZRT_TRI_FLOATS;
        double m_t = 0.0;
        for (int v = 0; v < 9; v++) m_t = __builtin_fmax(m_t, (double)(t[v] < 0 ? -t[v] : t[v]));
        mts.push_back(m_t);
        if (eta[(size_t)i] > 0.0 && eta[(size_t)i] < dinf) all.push_back(eta[(size_t)i]);
        if (eta[(size_t)i] <= cutoff) a_glob = __builtin_fmax(a_glob, eta[(size_t)i]);
      }
      std::nth_element(mts.begin(), mts.begin() + mts.size() / 2, mts.end());
      const double scale = mts[mts.size() / 2];
      if (a_glob > scale / 16384.0 && all.size() >= 2048) {
        const size_t k = all.size() - 1 - all.size() / 1024;
        std::nth_element(all.begin(), all.begin() + k, all.end());
        cutoff = __builtin_fmin(cutoff, all[k]);
      }
    }
  lap("  cutoff quantile");
    double a_max = 0.0;
    std::vector<double> fin;
    for (int i = 0; i < n_tri; i++) {
      if (eta[(size_t)i] <= cutoff) {
        a_max = __builtin_fmax(a_max, eta[(size_t)i]);
        if (eta[(size_t)i] > 0.0) fin.push_back(eta[(size_t)i]);
      } else if (eta[(size_t)i] < dinf) {
        prune_bad++; // (a bound, but a useless one)
      }
    }
    if (!fin.empty()) {
      std::nth_element(fin.begin(), fin.begin() + fin.size() / 2, fin.end());
      prune_A_med = 2.0 * fin[fin.size() / 2];
    }
    prune_a = __builtin_nextafterf((float)(2.0 * a_max), __builtin_inff());
  lap("  a_max");
    std::vector<unsigned char> node_flag(tree4.size(), 0); // (ids are topologically ordered: children after parents)
    for (int i = (int)tree4.size() - 1; i >= 1; i--) {
      const HostNode& h = tree4[(size_t)i];
      unsigned char f = 0;
      if (h.n > 0) {
        for (int k = h.index; k < h.index + h.n; k++) f |= eta[(size_t)k] > cutoff;
      } else {
        f = node_flag[(size_t)h.left] | node_flag[(size_t)h.right];
      }
      node_flag[(size_t)i] = f;
    }
    // REF_NOPRUNE on every reference to a record with such a triangle below it (and on the root reference)
    for (size_t q = 0; q < rec_slot_nodes.size(); q++) {
      uint32_t rf[4];
      memcpy(rf, &inner4[q * N4_FLOAT4 + N4_ROW_REF], sizeof rf);
      for (int k = 0; k < 4; k++) {
        const int nd = rec_slot_nodes[q][k];
        if (nd > 0 && (int32_t)rf[k] >= 0 && node_flag[(size_t)nd]) {
          rf[k] |= REF_NOPRUNE;
          prune_flagged++;
        }
      }
      memcpy(&inner4[q * N4_FLOAT4 + N4_ROW_REF], rf, sizeof rf);
    }
    root4_flag = node_flag[1] ? REF_NOPRUNE : 0u;
    // the same flags per child in the BINARY records (the in-order kernel prunes too: ezrt_traceq.h), over the caller's tree
    {
      std::vector<unsigned char> rflag((size_t)n_nodes, 0);
      for (int i = n_nodes - 1; i >= 1; i--) {
        const HostNode h = decode_node(nodes, i);
        unsigned char f = 0;
        if (h.n > 0) {
          for (int k = h.index; k < h.index + h.n; k++) f |= eta[(size_t)k] > cutoff;
        } else {
          f = rflag[(size_t)h.left] | rflag[(size_t)h.right];
        }
        rflag[(size_t)i] = f;
      }
      for (int i = 1; i < n_nodes; i++) {
        if (inner_id[(size_t)i] < 0) continue;
        const HostNode h = decode_node(nodes, i);
        float4& q3 = inner[(size_t)inner_id[(size_t)i] * 4 + 3];
        const uint32_t fl = rflag[(size_t)h.left] ? 1u : 0u, fr = rflag[(size_t)h.right] ? 1u : 0u;
        memcpy(&q3.z, &fl, 4);
        memcpy(&q3.w, &fr, 4);
      }
    }
    if (root4_flag) prune_flagged++;
  }

  lap("pruning bounds + flags");
  // ---- shading records + table of distinct materials (bitwise distinct 18-float tuples)
  std::vector<float4> shade((size_t)n_tri * SHADE_REC_FLOAT4), mats;
  {
    std::map<std::array<uint32_t, 18>, uint32_t> index;
    std::vector<uint32_t> mat_id((size_t)n_tri);
    std::array<uint32_t, 18> last_key;
    uint32_t last_id = 0;
    for (int i = 0; i < n_tri; i++) { // (sequential: material numbers follow first appearance; consecutive triangles mostly share one)
      const float* t = tri + (size_t)i * EZRT_TRI_FLOATS;
      std::array<uint32_t, 18> key;
      memcpy(key.data(), t + 18, sizeof(uint32_t) * 18);
      if (i > 0 && key == last_key) {
        mat_id[(size_t)i] = last_id;
        continue;
      }
      auto it = index.find(key);
      if (it == index.end()) {
        it = index.emplace(key, (uint32_t)index.size()).first;
        Mat m;
        m.emissive = f3{t[18], t[19], t[20]};
        m.baseColor = f3{t[21], t[22], t[23]};
        m.subsurface = t[24];
        m.metallic = t[25];
        m.specular = t[26];
        m.specularTint = t[27];
        m.roughness = t[28];
        m.anisotropic = t[29];
        m.sheen = t[30];
        m.sheenTint = t[31];
        m.clearcoat = t[32];
        m.clearcoatGloss = t[33];
        mat_derive(m);
        mats.push_back(make_float4(t[18], t[19], t[20], t[21]));
        mats.push_back(make_float4(t[22], t[23], t[24], t[25]));
        mats.push_back(make_float4(t[26], t[27], t[28], t[29]));
        mats.push_back(make_float4(t[30], t[31], t[32], t[33]));
        mats.push_back(make_float4(t[34], t[35], m.Cspec0.x, m.Cspec0.y));
        mats.push_back(make_float4(m.Cspec0.z, m.Csheen.x, m.Csheen.y, m.Csheen.z));
        mats.push_back(make_float4(m.alpha_gtr2, m.alpha_gtr1, m.gtr1_a2m1, m.gtr1_pilog));
      }
      last_key = key;
      last_id = it->second;
      mat_id[(size_t)i] = last_id;
    }
    parallel_for(n_tri, 1 << 15, [&](int lo_i, int hi_i, int) {
    for (int i = lo_i; i < hi_i; i++) {
      const float* t = tri + (size_t)i * EZRT_TRI_FLOATS;
      const ShadeDen dn = shade_denominators(f3{t[0], t[1], t[2]}, f3{t[3], t[4], t[5]}, f3{t[6], t[7], t[8]});
      float mi;
      const uint32_t mu = mat_id[(size_t)i];
      memcpy(&mi, &mu, 4);
      float4* o = &shade[(size_t)i * SHADE_REC_FLOAT4];
      o[0] = make_float4(t[9], t[10], t[11], t[12]);
      o[1] = make_float4(t[13], t[14], t[15], t[16]);
      o[2] = make_float4(t[17], mi, 0.0f, 0.0f);
      o[3] = make_float4(dn.a5, dn.b5, dn.a34, dn.b34);
    }
    });
  }

  lap("shading records");
  EzrtScene* s = new (std::nothrow) EzrtScene();
  if (!s) return fail(EZRT_ERR_NOMEM, "out of memory");
  s->n_tri = n_tri;
  s->n_materials = (int)(mats.size() / MAT_REC_FLOAT4);
  s->n_nodes = n_nodes;
  s->depth = maxd;
  s->n_inner = n_inner;
  s->root_ref = ref_of(1);
#define SC_TRY(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      delete s;                                                                                   \
      return fail(EZRT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));                \
    }                                                                                             \
  } while (0)
  SC_TRY(s->tri_geom.ensure(geom.size()));
  SC_TRY(s->tri_ref.ensure((size_t)n_tri * EZRT_TRI_FLOATS));
  SC_TRY(s->inner.ensure(inner.size()));
  SC_TRY(s->counters.ensure((size_t)CTR_SLOTS * EZRT_CTR_COUNT));
  SC_TRY(hipMemcpy(s->tri_geom.p, geom.data(), geom.size() * sizeof(float4), hipMemcpyHostToDevice));
  SC_TRY(s->tri_shade.ensure(shade.size()));
  SC_TRY(s->mat_table.ensure(mats.size()));
  SC_TRY(hipMemcpy(s->tri_shade.p, shade.data(), shade.size() * sizeof(float4), hipMemcpyHostToDevice));
  SC_TRY(hipMemcpy(s->mat_table.p, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
  SC_TRY(hipMemcpy(s->tri_ref.p, tri, (size_t)n_tri * EZRT_TRI_FLOATS * sizeof(float), hipMemcpyHostToDevice));
  SC_TRY(hipMemcpy(s->inner.p, inner.data(), inner.size() * sizeof(float4), hipMemcpyHostToDevice));
  s->n_inner4 = n_inner4;
  s->stack_need4 = stack_need4;
  s->retreed = retreed;
  s->prunable = prunable;
  s->prune_G = prune_G;
  s->prune_Z = prune_Z;
  s->prune_M = prune_M;
  s->prune_A_med = prune_A_med;
  s->prune_a = prune_a;
  s->prune_bad = prune_bad;
  s->prune_flagged = prune_flagged;
  s->root4 = n_inner4 > 0 ? root4_flag : s->root_ref;
  if (!tri_leaf_h.empty()) {
    SC_TRY(s->tri_leaf.ensure(tri_leaf_h.size()));
    SC_TRY(s->ref_up.ensure(ref_up_h.size()));
    SC_TRY(hipMemcpy(s->tri_leaf.p, tri_leaf_h.data(), tri_leaf_h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    SC_TRY(hipMemcpy(s->ref_up.p, ref_up_h.data(), ref_up_h.size() * sizeof(int2), hipMemcpyHostToDevice));
  }
  if (n_inner4 > 0) {
    SC_TRY(s->inner4.ensure(inner4.size()));
    SC_TRY(hipMemcpy(s->inner4.p, inner4.data(), inner4.size() * sizeof(float4), hipMemcpyHostToDevice));
  }
  SC_TRY(hipMemset(s->counters.p, 0, (size_t)CTR_SLOTS * EZRT_CTR_COUNT * sizeof(unsigned long long)));
#undef SC_TRY
  lap("device allocation + upload");
  s->stats[0] = n_tri;
  s->stats[1] = n_nodes;
  s->stats[2] = maxd;
  s->stats[3] = leaves;
  s->stats[4] = maxleaf;
  s->stats[5] = (int64_t)(geom.size() * sizeof(float4) + (size_t)n_tri * 144 + inner.size() * sizeof(float4) +
                          inner4.size() * sizeof(float4) + shade.size() * sizeof(float4) + mats.size() * sizeof(float4));
  *out = s;
  return 0;
}

static int ezrt_scene_set_env_body(EzrtScene* s, const float* hdr, const float* cache, int w, int h, int filter) {
  if (!s || !hdr || w <= 0 || h <= 0) return fail(EZRT_ERR_INVALID, "bad env arguments");
  if (filter != EZRT_FILTER_NEAREST && filter != EZRT_FILTER_BILINEAR) return fail(EZRT_ERR_INVALID, "bad filter");
  if ((int64_t)w * w / 2 >= ((int64_t)1 << 31)) return fail(EZRT_ERR_UNSUPPORTED, "hdrResolution^2/2 overflows int");
  size_t n = (size_t)w * h;
  std::vector<float4> tmp(n);
  for (size_t i = 0; i < n; i++) tmp[i] = make_float4(hdr[i * 3], hdr[i * 3 + 1], hdr[i * 3 + 2], 0.0f);
  HIP_TRY(s->hdr.ensure(n));
  HIP_TRY(hipMemcpy(s->hdr.p, tmp.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  // RGBE form: every texel exactly (m / 256) * 2^(E - 128) per channel with one shared E (what HDRLoader produces)
  s->has_rgbe = false;
  {
    std::vector<uint32_t> packed(n);
    bool ok = true;
    for (size_t i = 0; i < n && ok; i++) {
      const float c[3] = {hdr[i * 3], hdr[i * 3 + 1], hdr[i * 3 + 2]};
      uint32_t bits[3];
      memcpy(bits, c, sizeof bits);
      if ((bits[0] | bits[1] | bits[2]) == 0u) { // +0 +0 +0
        packed[i] = 0u;
        continue;
      }
      float mx = c[0] > c[1] ? c[0] : c[1];
      mx = mx > c[2] ? mx : c[2];
      if (!(mx > 0.0f) || !(c[0] >= 0.0f) || !(c[1] >= 0.0f) || !(c[2] >= 0.0f) || mx > 3.0e38f) { // negative, NaN, inf, -0
        ok = false;
        break;
      }
      int k = 0;
      (void)frexpf(mx, &k); // mx = f * 2^k, f in [0.5, 1)
      const int E = k + 128;
      if (E < 0 || E > 255) {
        ok = false;
        break;
      }
      uint32_t m[3];
      for (int j = 0; j < 3 && ok; j++) {
        if (bits[j] == 0x80000000u) ok = false; // -0 would decode as +0
        const float q = ldexpf(c[j], 8 - k); // exact scaling
        const uint32_t mi = (uint32_t)q;
        if (!(q >= 0.0f && q < 256.0f) || (float)mi != q || ldexpf((float)mi, E - 136) != c[j]) ok = false;
        m[j] = mi;
      }
      if (ok) packed[i] = m[0] | (m[1] << 8) | (m[2] << 16) | ((uint32_t)E << 24);
    }
    if (ok) {
      HIP_TRY(s->hdr_rgbe.ensure(n));
      HIP_TRY(hipMemcpy(s->hdr_rgbe.p, packed.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
      s->has_rgbe = true;
    }
  }
  s->has_cache = false;
  if (cache) {
    for (size_t i = 0; i < n; i++) tmp[i] = make_float4(cache[i * 3], cache[i * 3 + 1], cache[i * 3 + 2], 0.0f);
    HIP_TRY(s->cache.ensure(n));
    HIP_TRY(hipMemcpy(s->cache.p, tmp.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    {
      std::vector<float2> xy(n);
      std::vector<float> pdf(n);
      for (size_t i = 0; i < n; i++) {
        xy[i] = make_float2(cache[i * 3], cache[i * 3 + 1]);
        pdf[i] = cache[i * 3 + 2];
      }
      HIP_TRY(s->cache_xy.ensure(n));
      HIP_TRY(s->cache_pdf.ensure(n));
      HIP_TRY(hipMemcpy(s->cache_xy.p, xy.data(), n * sizeof(float2), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->cache_pdf.p, pdf.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    s->has_cache = true;
  }
  s->env_w = w;
  s->env_h = h;
  s->env_filter = filter;
  return 0;
}
int ezrt_scene_set_env(EzrtScene* s, const float* hdr, const float* cache, int w, int h, int filter) {
  return ezi::guarded("ezrt_scene_set_env", [&]() -> int { return ezrt_scene_set_env_body(s, hdr, cache, w, h, filter); });
}

} // extern "C"
