// ezrt_hip.hip -- libezrt_hip.so: the C ABI of include/ezrt.h implemented on
// hand-written gfx950 kernels (ezrt_kernels.h).  Host code here only validates,
// re-lays the scene out for the GPU, and launches; there is no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <exception>
#include <new>
#include <system_error>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "ezrt.h"
#include "ezrt_kernels.h"
#include "ezrt_wavefront.h"
#include "ezrt_traceq4.h"
#include "ezrt_streams.h"

using namespace ezd;

namespace {

thread_local char g_err[512];
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(EZRT_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
};

constexpr int MAX_TRACE_EVENTS = 2048; // launch_events: pairs of timing events a call can record (C5 at 512 spp: 576 trace launches);
                                       // the first 64 are created with the scene's other events, the rest when a call first needs them

// Schedule knobs (never change results).  Defaults = the tuned values for C2 on MI355X; an
// environment variable EZRT_<NAME> overrides the default at scene creation, ezrt_set_option at run time.
struct Tuning {
  int megakernel = 0;      // 1: v1 one-lane-per-path kernel instead of the streaming pipeline
  int leaf_threshold = 12; // lanes waiting at a leaf that trigger the triangle phase (24 until the traversal pruned: 12 is +3 % on C3 / C5 now)
  int pool_div = 1;        // ray-index pool = clamp(n_rays / (n_waves * pool_div), 8, pool_max)
  int pool_min = 8;        // fewest rays a wave is dealt (clamped to pool_max): a short queue then goes to fewer, fuller waves
  int pool_max = 128;      // rays per dynamic reservation (two 8x8 sub-blocks; affordable since the reservations spread over 8 counters)
  int trace_wps_rel = 7;   // waves per SIMD of the primary stage's launch (traceq4_kernel<.., true>; 0: trace_wps).  That variant
                           // needs 79 VGPRs; at 72 it spills 7 and is still 1.5 % faster (seven waves hide more latency)
  int trace_wps = 6;       // traceq_kernel register budget: waves per SIMD (4, 5, 6 or 8); 6 = 80 VGPRs (8 B of
                           // scratch) and fewer tree records in LDS, still +3.6 % over 5 since the loop got leaner
  int lds_nodes = 1 << 20; // cap on top-of-tree records staged in LDS
  int scatter = 1;         // primary rays enter the queue in a scattered 8x8 sub-block order (balances pools)
  int static_pct = 50;     // share of a trace queue dealt statically to the waves, percent (0: one pool each)
  int refill_min = 24;     // free lanes a wave waits for before it runs its refill code (publish results, adopt the
                           // prefetched ray, prefetch the next): wave-wide code for per-lane events, so batch it (+9 %)
  int lazy_dir = 1;        // the primary stage's first shading pass reads a ray's direction only after its hit record said "miss"
  int refill_min_rel = 40; // ... of the primary stage's launch (rays with a common origin; 0: refill_min): its refill also generates the rays
                           // (C2 +1.8 %, C4 +3.1 %, C3 / C5 +0.3 % over 24)
  int rel_boxes = 1;       // primary rays traverse boxes already translated by the eye
  int steal = 1;           // intra-wave work stealing in traceq_kernel (+ a redo launch for exact ties)
  int wide4 = 1;           // traceq4_kernel (4-wide collapse of the tree, ezrt_traceq4.h) for the timed stages; 0: the
                           // binary traceq_kernel.  Instrumented runs (level 1) and scenes whose boxes are not
                           // nested always use the binary kernel.
  int debug_stages = 0;    // print per-stage queue sizes (synchronises)
  int redo_overlap = 0;    // 1: the redo launch of a stage (exact ties beyond two candidates, rays that are not tame, stack overflows) runs on a
                           // side stream under the stage's first shading pass and the second pass waits for it; 0 (default since round 3):
                           // in line, before any shading.  The lists are EMPTY on the BASELINE configs since ties and zero-component rays
                           // stay in the 4-wide kernel, an empty launch is ~4 us, and the two measure the same (C2 / C4 / C5 within 0.3 %)
                           // -- without a cross-stream event wait in every frame (ezrt_streams.h: those can enter a slow state)
  int debug_oom_above = 0;     // test hook: chunk scratch for more than this many pixel-samples is reported as out of memory (exercises the
                               // smaller-chunk retry of ezrt_render_device)
  int debug_force_pending = 0; // test hook: every k-th ray slot takes the not-tame route (HIT_PENDING -> redo -> second pass)
  int shade_wgs = 0;       // workgroups of a shading launch, each looping over its share of the queue (0: 12 per CU = three full
                           // rounds of the first pass at 4 workgroups per CU and four of the second at 3; 4096 left the second
                           // pass with a third of a round at its end: -1.2 % on C2)
  int chunk_log2 = 26;     // pixel-samples in flight per chunk of a call, log2.  The small late stages of a chunk are latency-
                           // bound (their length is the deepest ray's, not their work), so bigger chunks amortise them: 2^24 ->
                           // 2^26 is +9 % on C4 (64 spp calls), +15 % on C5, +17 % on 256-spp C2 calls; 2^28 another 3-5 %.
                           // Scratch is sized by the call (<= ~350 B per pixel-sample in flight: 23 GB of the 288 at 2^26)
  int launch_events = 0;   // 1: a pair of timing events around every trace launch (ezrt_last_render_ms's second figure; each
                           // record costs the stream ~5 us: -1.2 % on C2); 0: only the call's begin / end events
  int env_planes = 1;      // the env cache as an (x, y) plane and a pdf plane for bilinear lookups (one load per row)
  int env_rgbe = 1;        // environment lookups through the 4-byte RGBE form of the map when it has an exact one (set_env)
  int min_staged = 16;      // a trace launch gives up workgroups per CU (down to 4) until this many top-of-tree records fit in LDS
  int rel_min_records = 4;  // the primary stage runs at trace_wps_rel waves per SIMD only while that leaves this many top-of-tree records in LDS
                            // (24 until round 4; C2 and C4 -- 16 stack rows, 6 records left at 7 workgroups per CU -- gain 0.6-1.1 % at 7)
  int gen_primary = 1;     // primary rays are generated inside the primary stage's trace and shading kernels (primary_dir) instead of
                           // written to a queue by raygen_kernel (timed pipeline with the 4-wide, eye-relative records only)
  int anyhit = 1;          // env shadow rays (the even slots of the MIS integrators' bounce stages) stop at their first accepted hit:
                           // the shading stage only asks whether they hit anything (never in the audit routes)
  int semi = 1;            // rays with an exactly-zero direction component: 1 (default) traversed by the 4-wide kernel in the launches that
                           // see them in numbers (the MIS integrators' bounce stages: SampleHdr's directions), 0 always the redo list
                           // (in-order kernel, one lane per ray), 2 in every launch without a common origin
  int tie_lca = 1;         // exact ties of the 4-wide kernel are ordered in place at the two leaves' lowest common ancestor in the
                           // reference's tree (tie_precedes, ezrt_traceq4.h); 0: every tie goes to the redo list
  int retree = 1;          // READ AT SCENE CREATION (EZRT_RETREE): the 4-wide records are built over a binned-SAH tree of the
                           // reference's LEAVES instead of over a cut of the reference's own inner nodes (retree_leaves below)
  int prune = 2;           // traceq4_kernel's distance pruning (ezrt_traceq4.h "Distance pruning": proven results-neutral): 0 the
                           // reference's unpruned traversal, 1 skip slots provably beyond the best hit, 2 that + nearest slot first
  int prune_mis = 2;       // ... of the MIS integrators' bounce stages (two rays per path, one an env shadow ray) when prune == 2:
                           // slot order there (1) was a wash on C4 (14.36 vs 14.28 Grays/s) and lost 6 % on C5 (2.46 vs 2.61)
  int prune_min_records = 0; // scenes with fewer 4-wide records than this are traced unpruned (small trees gain nothing)
  int stack_cap = 0;       // prune 2: LDS stack rows of the nearest-first traversal before a ray is handed to the redo list (0: the exact
                           // worst case of the slot-order traversal).  Fewer rows = more top-of-tree records staged in LDS
  int debug_stack_cap = 0; // test hook (prune 2): > 0 = stack rows beyond which a ray goes to the redo list, instead of the scene's bound
  int bounce_scatter = 1;  // the bounce stages' trace launches draw their queue in a scattered order, in granules of 8 rays
                           // (TraceQ4Args::gscat_shift; 0: consecutive slots, the order the shading stage wrote; 1: queues with one
                           // ray per path.  The MIS integrators' two-ray queues lost 1.3-2.8 % with it: never scattered)
  int pipeline_calls = 1;  // consecutive chunks -- of one call or of consecutive calls -- alternate between the two scratch sets and their own
                           // streams, so that a chunk's latency-bound late stages run under the next chunk's primary stage; only the
                           // accumulation into the caller's frame buffer stays on the caller's stream, in order (ezrt_render_device).
                           // 1 (default; 2 is accepted as the same): every scene; 0: never
  int static_pct_pipelined = 0; // static_pct of the trace launches of a pipelined chunk: its workgroups become resident as the other chunk's
                           // launches free wave slots, and a pool dealt statically to a workgroup that arrives late is the launch's tail.
                           // With the queues all dynamic pipelining gains on every config (C3 +3.0 %, C4 +2.6 %, C5 +1.8 %, C2 +13 %); with
                           // the unpipelined optimum of 50 it lost 1-5 % on C3 / C4 / C5 (profiles/r4/pipeline_calls_ab.txt)
  int handover = 1;        // traceq4_kernel: once the queue is exhausted, idle lanes take the prefetched (unstarted) rays of lanes of their wave that
                           // are still traversing (TraceQ4Args::handover)
  int steal_bound = 1;     // traceq4_kernel: a lane that takes a pending subtree of another lane's ray prunes against that lane's best hit so far
  int audit_via_queue = 0; // 1: ezrt_query_hits and ezrt_render_paths run through the TIMED kernels (traceq_kernel with
                           // the template, LDS layout, stealing and redo launch of a render call + the streaming shading
                           // stages) instead of the in-order audit kernels; 2 (query only): additionally treat the rays as
                           // sharing ray 0's origin, i.e. the primary stage's const_origin / pre-translated-box variant
};
struct TuningName {
  const char* name;
  int Tuning::*field;
  int lo, hi; // accepted range (ezrt_set_option rejects anything else; environment overrides are clamped)
};
const TuningName kTuning[] = {{"megakernel", &Tuning::megakernel, 0, 1},
                              {"leaf_threshold", &Tuning::leaf_threshold, 1, 64},
                              {"pool_div", &Tuning::pool_div, 1, 1 << 16},
                              {"pool_max", &Tuning::pool_max, 8, 4096},
                              {"pool_min", &Tuning::pool_min, 8, 4096},
                              {"trace_wps", &Tuning::trace_wps, 1, 8},
                              {"trace_wps_rel", &Tuning::trace_wps_rel, 0, 8},
                              {"lds_nodes", &Tuning::lds_nodes, 0, 1 << 24},
                              {"steal", &Tuning::steal, 0, 1},
                              {"rel_boxes", &Tuning::rel_boxes, 0, 1},
                              {"refill_min", &Tuning::refill_min, 1, 64},
                              {"refill_min_rel", &Tuning::refill_min_rel, 0, 64},
                              {"lazy_dir", &Tuning::lazy_dir, 0, 1},
                              {"static_pct", &Tuning::static_pct, 0, 95},
                              {"scatter", &Tuning::scatter, 0, 8},
                              {"wide4", &Tuning::wide4, 0, 1},
                              {"debug_stages", &Tuning::debug_stages, 0, 2},
                              {"env_rgbe", &Tuning::env_rgbe, 0, 1},
                              {"env_planes", &Tuning::env_planes, 0, 1},
                              {"launch_events", &Tuning::launch_events, 0, 1},
                              {"shade_wgs", &Tuning::shade_wgs, 0, 4096},
                              {"chunk_log2", &Tuning::chunk_log2, 12, 28},
                              {"redo_overlap", &Tuning::redo_overlap, 0, 1},
                              {"debug_force_pending", &Tuning::debug_force_pending, 0, 1 << 20},
                              {"debug_oom_above", &Tuning::debug_oom_above, 0, 1 << 30},
                              {"gen_primary", &Tuning::gen_primary, 0, 1},
                              {"rel_min_records", &Tuning::rel_min_records, 0, 4096},
                              {"min_staged", &Tuning::min_staged, 0, 4096},
                              {"anyhit", &Tuning::anyhit, 0, 1},
                              {"semi", &Tuning::semi, 0, 2},
                              {"tie_lca", &Tuning::tie_lca, 0, 1},
                              {"retree", &Tuning::retree, 0, 1},
                              {"prune", &Tuning::prune, 0, 2},
                              {"prune_mis", &Tuning::prune_mis, 0, 2},
                              {"prune_min_records", &Tuning::prune_min_records, 0, 1 << 24},
                              {"stack_cap", &Tuning::stack_cap, 0, 64},
                              {"debug_stack_cap", &Tuning::debug_stack_cap, 0, 64},
                              {"bounce_scatter", &Tuning::bounce_scatter, 0, 1},
                              {"pipeline_calls", &Tuning::pipeline_calls, 0, 2},
                              {"static_pct_pipelined", &Tuning::static_pct_pipelined, 0, 95},
                              {"handover", &Tuning::handover, 0, 1},
                              {"steal_bound", &Tuning::steal_bound, 0, 1},
                              {"audit_via_queue", &Tuning::audit_via_queue, 0, 2}};
Tuning tuning_from_env() {
  Tuning t;
  for (const TuningName& k : kTuning) {
    std::string env = "EZRT_";
    for (const char* c = k.name; *c; c++) env += (char)((*c >= 'a' && *c <= 'z') ? (*c - 32) : *c);
    if (const char* e = getenv(env.c_str())) {
      int v = atoi(e);
      t.*(k.field) = v < k.lo ? k.lo : (v > k.hi ? k.hi : v);
    }
  }
  return t;
}

} // namespace

// Scratch of one sub-chunk of frames in flight (see EzrtScene::pipe).
struct Pipe {
  DevBuf<Sample3> samples;
  // wavefront queues (ping-pong)
  DevBuf<float4> rq_o[2], rq_d[2];
  DevBuf<float4> st[2][5];
  DevBuf<int2> hits2[2];        // hit records, ping-pong with the ray queues
  DevBuf<uint32_t> redo_flag;   // per ray slot: already on the redo list
  DevBuf<unsigned long long> wave_log; // debug_stages=2 only
  DevBuf<float> sobol_tab;  // [frames of the chunk][8]
  DevBuf<uint32_t> qcounts; // [0..63] path counts per stage, [64..99] trace queue heads, [100..115] debug,
                            // [120] redo count, [121] redo queue head
  DevBuf<uint32_t> redo_slots;
  DevBuf<uint32_t> qheads;      // traceq reservation counters: [launch slot][TRACE_HEADS][TRACE_HEAD_STRIDE] (QHEADS_WORDS in all; zeroed per chunk)
  DevBuf<float4> inner_rel;     // inner records translated by -eye (primary rays)
  DevBuf<float4> inner4_rel;    // 4-wide records translated by -eye
  DevBuf<uint4> defer_list;     // split shading: paths with a surface interaction, per workgroup
  DevBuf<uint32_t> defer_count;
  int stream_device = 0;         // device `stream` and `side` belong to (they return to its pool)
  hipStream_t side = nullptr;    // redo launches that overlap the first shading pass
  hipEvent_t ev_main = nullptr;  // a stage's main trace launch is enqueued / done
  hipEvent_t ev_redo = nullptr;  // ... its redo launch is done
  hipStream_t stream = nullptr;  // own stream (pipelined calls only)
  hipEvent_t ev_done = nullptr;  // samples of the sub-chunk are complete
  hipEvent_t ev_free = nullptr;  // ... and have been folded into the frame buffer
  bool free_recorded = false;    // ev_free has been recorded at least once (pipeline_calls: the next user of this scratch set waits for it)
};

struct EzrtScene {
  int n_tri = 0, n_nodes = 0;
  DevBuf<float4> tri_geom;
  DevBuf<float4> tri_shade, mat_table; // per-triangle shading records, distinct materials (ezrt_device.h: shade_point)
  int n_materials = 0;
  DevBuf<float> tri_ref;
  DevBuf<float4> inner;
  DevBuf<int32_t> tri_leaf;   // reference leaf node of every triangle; ref_up: per reference node (parent | depth << 24, parent's
  DevBuf<int2> ref_up;        // binary record | is-right-child << 31) -- tie_precedes (empty: ties go to the redo list)
  DevBuf<float4> inner4;      // 4-wide records (ezrt_traceq4.h), breadth-first; empty when the boxes are not nested
  int n_inner4 = 0;
  int stack_need4 = 1;        // LDS stack rows the 4-wide traversal can need (exact worst case over hit patterns)
  // distance pruning (ezrt_traceq4.h): scene maxima of the per-triangle bound, evaluated in double at create
  bool retreed = false;       // the 4-wide records are a collapse of retree_leaves' tree, not of the caller's inner nodes
  bool prunable = false;      // every leaf box holds its triangles (and the boxes are nested: the 4-wide records exist)
  double prune_G = 0.0;       // max 1 / sin(theta'/2) over the triangles with a bound (diagnostic)
  double prune_Z = 0.0;       // max distance of a vertex from its triangle's stored plane (diagnostic)
  double prune_M = 0.0;       // max |coordinate| (diagnostic)
  float prune_a = 0.0f;       // launch argument: 2 max eta_T over the ordinary triangles, rounded up
  double prune_A_med = 0.0;   // 2 median eta_T (diagnostic)
  int64_t prune_bad = 0;      // triangles that are not ordinary (a large bound or none): the records above them are never pruned
  int64_t prune_flagged = 0;  // ... how many records that is
  uint32_t root4 = 0;
  DevBuf<float4> hdr, cache;
  DevBuf<float2> cache_xy; // the cache as two planes (bilinear lookups: one load per row; knob env_planes)
  DevBuf<float> cache_pdf;
  DevBuf<uint32_t> hdr_rgbe; // RGBE form of hdr (has_rgbe)
  bool has_rgbe = false;
  uint32_t root_ref = 0;
  int env_w = 0, env_h = 0, env_filter = 0;
  bool has_cache = false;
  uint32_t sobol_mask = 7u; // ezrt_scene_set_sampler
  int instr = 0;
  int depth = 0;
  int64_t stats[6] = {0, 0, 0, 0, 0, 0};
  DevBuf<unsigned long long> counters;
  // render scratch
  DevBuf<int2> blocks;
  std::vector<int2> blocks_host;
  EzrtRenderParams blocks_for; // params the block list was built for
  bool blocks_valid = false;
  DevBuf<float4> accum_tmp;
  // Two independent sets of render scratch: a call's frames are cut into sub-chunks that alternate
  // between them, each on its own stream, so one sub-chunk's latency-bound phases (the ends of the
  // persistent trace launches, the late bounces, launch gaps) run under the other's bulk work.
  Pipe pipe[ezh::SHARED_STREAMS]; // (two: deeper pipelines were measured in round 5 and removed in round 6)
  int num_cus = 0;
  uint32_t chunk_seq = 0;     // chunks rendered so far (pipeline_calls: chunk i uses scratch set i & 1)
  bool chunk_pipelined = false; // the chunk being enqueued runs on a scratch set's own stream (set by ezrt_render_device)
  int n_inner = 0;
  Tuning tune = tuning_from_env();
  // timing
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_trace[MAX_TRACE_EVENTS][2] = {};
  int n_trace_events = 0, n_trace_launches = 0, n_trace_events_created = 0;
  bool events_ready = false; // ensure_events ran to its end
  bool timed = false;

  DevScene dev() const {
    DevScene d;
    d.tri_geom = tri_geom.p;
    d.tri_ref = tri_ref.p;
    d.tri_shade = tri_shade.p;
    d.mat_table = mat_table.p;
    d.inner = inner.p;
    d.root_ref = root_ref;
    d.n_tri = n_tri;
    d.hdr = hdr.p;
    d.hdr_rgbe = (has_rgbe && tune.env_rgbe) ? hdr_rgbe.p : nullptr;
    d.cache = has_cache ? cache.p : nullptr;
    d.cache_xy = (has_cache && tune.env_planes && env_w >= 2) ? cache_xy.p : nullptr;
    d.cache_pdf = (has_cache && tune.env_planes && env_w >= 2) ? cache_pdf.p : nullptr;
    d.env_w = env_w;
    d.env_h = env_h;
    d.env_filter = env_filter;
    d.sobol_mask = sobol_mask;
    return d;
  }
};

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

int validate_params(const EzrtScene* s, const EzrtRenderParams* p) {
  if (!p) return fail(EZRT_ERR_INVALID, "params is NULL");
  if (p->width <= 0 || p->height <= 0) return fail(EZRT_ERR_INVALID, "width/height must be positive");
  if (p->x0 < 0 || p->y0 < 0 || p->x1 > p->width || p->y1 > p->height || p->x0 > p->x1 || p->y0 > p->y1)
    return fail(EZRT_ERR_INVALID, "pixel rect outside the image");
  if (p->max_bounce < 0 || p->max_bounce > 64) return fail(EZRT_ERR_INVALID, "max_bounce out of range [0,64]");
  if (p->integrator != 3 && p->integrator != 4 && p->integrator != 50 && p->integrator != 51 && p->integrator != 52)
    return fail(EZRT_ERR_INVALID, "unknown integrator");
  if (p->shard_count < 0 || p->shard_index < 0 || (p->shard_count > 0 && p->shard_index >= p->shard_count))
    return fail(EZRT_ERR_INVALID, "bad shard index/count");
  if (p->tile_w < 0 || p->tile_h < 0) return fail(EZRT_ERR_INVALID, "bad tile size");
  if ((p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO) && !s->has_cache)
    return fail(EZRT_ERR_INVALID, "integrator 51 needs the env cache (ezrt_scene_set_env)");
  return 0;
}

bool same_blocks(const EzrtRenderParams& a, const EzrtRenderParams& b) {
  return a.width == b.width && a.height == b.height && a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1 &&
         a.tile_w == b.tile_w && a.tile_h == b.tile_h && a.shard_index == b.shard_index && a.shard_count == b.shard_count;
}
// list of 16x16 pixel blocks holding at least one owned pixel: a block is kept iff one of the tiles that overlap
// (block AND rect) belongs to this shard -- a handful of tile cells per block instead of its 256 pixels
int build_blocks(EzrtScene* s, const EzrtRenderParams& p, hipStream_t st) {
  if (s->blocks_valid && same_blocks(s->blocks_for, p)) return 0;
  std::vector<int2>& v = s->blocks_host;
  v.clear();
  const int tw = p.tile_w > 0 ? p.tile_w : 32, th = p.tile_h > 0 ? p.tile_h : 32;
  const int tiles_x = (p.width + tw - 1) / tw;
  for (int by = (p.y0 / 16) * 16; by < p.y1; by += 16)
    for (int bx = (p.x0 / 16) * 16; bx < p.x1; bx += 16) {
      const int xa = std::max(bx, p.x0), xb = std::min(bx + 16, p.x1), ya = std::max(by, p.y0), yb = std::min(by + 16, p.y1);
      if (xa >= xb || ya >= yb) continue;
      bool any = p.shard_count <= 1;
      for (int ty = ya / th; ty <= (yb - 1) / th && !any; ty++)
        for (int tx = xa / tw; tx <= (xb - 1) / tw && !any; tx++) any = (ty * tiles_x + tx) % p.shard_count == p.shard_index;
      if (any) v.push_back(make_int2(bx, by));
    }
  if (!v.empty()) {
    HIP_TRY(s->blocks.ensure(v.size()));
    HIP_TRY(hipMemcpyAsync(s->blocks.p, v.data(), v.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st)); // v may be rebuilt by the next call
  }
  s->blocks_for = p;
  s->blocks_valid = true;
  return 0;
}

template <int INTEG>
void launch_trace_i(const TraceArgs& a, int mode, dim3 grid, size_t lds, hipStream_t st) {
  if (mode == 2) hipLaunchKernelGGL((trace_kernel<INTEG, false, true>), grid, dim3(BLOCK), lds, st, a);
  else if (mode == 1) hipLaunchKernelGGL((trace_kernel<INTEG, true, false>), grid, dim3(BLOCK), lds, st, a);
  else hipLaunchKernelGGL((trace_kernel<INTEG, false, false>), grid, dim3(BLOCK), lds, st, a);
}
// mode: 0 timed, 1 full counters, 2 path log
void launch_trace(const TraceArgs& a, int mode, dim3 grid, size_t lds, hipStream_t st) {
  switch (a.p.integrator) {
    case EZRT_INTEGRATOR_P3_DIFFUSE: launch_trace_i<EZRT_INTEGRATOR_P3_DIFFUSE>(a, mode, grid, lds, st); break;
    case EZRT_INTEGRATOR_P4_DISNEY: launch_trace_i<EZRT_INTEGRATOR_P4_DISNEY>(a, mode, grid, lds, st); break;
    case EZRT_INTEGRATOR_P5_SOBOL: launch_trace_i<EZRT_INTEGRATOR_P5_SOBOL>(a, mode, grid, lds, st); break;
    case EZRT_INTEGRATOR_P5_MIS_ANISO: launch_trace_i<EZRT_INTEGRATOR_P5_MIS_ANISO>(a, mode, grid, lds, st); break;
    default: launch_trace_i<EZRT_INTEGRATOR_P5_MIS>(a, mode, grid, lds, st); break;
  }
}

// Events and the shared stream pair of a scene, created on its first render call.  Idempotent and incremental: a
// step that fails leaves what exists in place -- counted, so that ezrt_scene_destroy releases it -- and the next call resumes
// there; `events_ready` is only set after the last step, so no call ever runs with a null stream or event.
int ensure_events(EzrtScene* s) {
  if (s->events_ready) return 0;
  if (!s->ev_begin) HIP_TRY(hipEventCreate(&s->ev_begin));
  if (!s->ev_end) HIP_TRY(hipEventCreate(&s->ev_end));
  while (s->n_trace_events_created < 64) {
    const int i = s->n_trace_events_created;
    if (!s->ev_trace[i][0]) HIP_TRY(hipEventCreate(&s->ev_trace[i][0]));
    if (!s->ev_trace[i][1]) HIP_TRY(hipEventCreate(&s->ev_trace[i][1]));
    s->n_trace_events_created = i + 1;
  }
  if (!s->pipe[0].stream) {
    // (the device's shared pair: ezrt_streams.h says why the two streams the chunks alternate between are not the scene's own)
    hipStream_t pair[ezh::SHARED_STREAMS];
    int dev = 0;
    HIP_TRY(ezh::stream_shared_pair(pair, &dev));
    for (int i = 0; i < ezh::SHARED_STREAMS; i++) {
      s->pipe[i].stream = pair[i];
      s->pipe[i].stream_device = dev;
    }
  }
  for (Pipe& q : s->pipe) {
    // (the side stream of the redo launches -- knob redo_overlap, off by default -- is taken from the pool when first needed)
    if (!q.ev_main) HIP_TRY(hipEventCreateWithFlags(&q.ev_main, hipEventDisableTiming));
    if (!q.ev_redo) HIP_TRY(hipEventCreateWithFlags(&q.ev_redo, hipEventDisableTiming));
    if (!q.ev_done) HIP_TRY(hipEventCreateWithFlags(&q.ev_done, hipEventDisableTiming));
    if (!q.ev_free) HIP_TRY(hipEventCreateWithFlags(&q.ev_free, hipEventDisableTiming));
  }
  s->events_ready = true;
  return 0;
}

size_t stack_lds_bytes(const EzrtScene* s) {
  int entries = s->depth > 1 ? s->depth : 1; // pending far children <= depth - 1
  return (size_t)entries * BLOCK * sizeof(int);
}



// Launch configuration of traceq_kernel (shared by the render pipeline and the audit routes).
// LDS per workgroup: traversal stack + lane table + as many top-of-tree records (80 B each) as fit
// when the register budget's `trace_wps` waves/SIMD (= trace_wps workgroups of 256 per CU) are resident
struct TraceCfg {
  size_t lds = 0, lds_t = 0;
  int lds_nodes = 0, blocks_per_cu = 1;
  unsigned grid_full = 0;
};
int ensure_num_cus(EzrtScene* s) {
  if (!s->num_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return 0;
}
TraceCfg trace_cfg(const EzrtScene* s) { // needs s->num_cus
  TraceCfg c;
  const Tuning& tu = s->tune;
  c.lds = stack_lds_bytes(s);
  const size_t lds_fixed = c.lds + BLOCK * sizeof(int);
  int blocks_per_cu = tu.trace_wps > 0 ? tu.trace_wps : 5;
  if ((size_t)blocks_per_cu * lds_fixed > 158 * 1024) blocks_per_cu = (int)((158 * 1024) / lds_fixed);
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  size_t lds_budget = (size_t)(158 * 1024) / blocks_per_cu;
  if (lds_budget > 64 * 1024) lds_budget = 64 * 1024; // static cap of a dynamic-LDS launch without opt-in
  lds_budget -= lds_budget / 16; // allocation-granule slack: a workgroup must not lose its CU slot to rounding
  int lds_nodes = lds_budget > lds_fixed ? (int)((lds_budget - lds_fixed) / 80) : 0;
  if (lds_nodes > s->n_inner) lds_nodes = s->n_inner;
  if (lds_nodes > tu.lds_nodes) lds_nodes = tu.lds_nodes;
  if (lds_nodes < 0) lds_nodes = 0;
  c.lds_nodes = lds_nodes;
  c.lds_t = lds_fixed + (size_t)lds_nodes * 80;
  c.blocks_per_cu = blocks_per_cu;
  c.grid_full = (unsigned)(s->num_cus * blocks_per_cu);
  return c;
}
// the template instance a render call uses for this scene's settings
void launch_traceq_cfg(EzrtScene* s, const TraceCfg& c, const TraceQArgs& q, bool small, hipStream_t st) {
  const unsigned trace_grid = small ? 64u : c.grid_full; // redo lists are (nearly) empty
  // (one register budget since round 6: 6 waves per SIMD = 80 VGPRs; knob trace_wps only sets the workgroups per CU)
  if (s->instr > 0) hipLaunchKernelGGL((traceq_kernel<true, 6>), dim3(trace_grid), dim3(BLOCK), c.lds_t, st, q);
  else hipLaunchKernelGGL((traceq_kernel<false, 6>), dim3(trace_grid), dim3(BLOCK), c.lds_t, st, q);
  s->n_trace_launches++;
}
// the same for traceq4_kernel: fewer stack rows (stack_need4), 112-B records in LDS
// rel: the launch traverses boxes translated by a common origin (traceq4_kernel<.., true>); the other variant keeps
// the ray directions in LDS (3 floats per lane after the lane table)
// waves per SIMD of a traceq4 launch: the primary stage's variant may run one more (trace_wps_rel), but only while
// that still leaves room for a useful top of the tree in LDS (deep trees need the space for stack rows: C5 and C3
// would stage ONE record at 7 workgroups per CU and lose 3 %)
// distance pruning of the timed stages: knob, scene property, instrumentation off
int prune_mode(const EzrtScene* s) {
  if (!s->prunable || s->n_inner4 < s->tune.prune_min_records) return 0;
  return s->tune.prune;
}
// stack rows of a traceq4 launch: the exact worst case of the slot-order traversal; the nearest-first order (prune 2)
// has no small bound -- it runs with the same rows as its cap (a ray beyond it goes to the redo list) + three rows of
// slack, because one step pushes up to three entries before the cap is tested
int stack_cap4(const EzrtScene* s) { // (prune 2 only: the other modes have no overflow route)
  const int c = s->tune.stack_cap;
  return (c > 0 && c < s->stack_need4) ? c : s->stack_need4;
}
int stack_rows4(const EzrtScene* s) { return prune_mode(s) == 2 ? stack_cap4(s) + 3 : s->stack_need4; }
int records_staged4(const EzrtScene* s, int wps) {
  const size_t lds_fixed = (size_t)stack_rows4(s) * BLOCK * sizeof(int) + BLOCK * sizeof(int);
  size_t budget = (size_t)(158 * 1024) / (size_t)(wps > 0 ? wps : 1);
  if (budget > 64 * 1024) budget = 64 * 1024;
  budget -= budget / 16;
  return budget > lds_fixed ? (int)((budget - lds_fixed) / (N4_LDS_DWORDS * 4)) : 0;
}
int wps4(const EzrtScene* s, bool rel) {
  const int w = s->tune.trace_wps_rel;
  if (rel && w > 0 && (w <= s->tune.trace_wps || records_staged4(s, w) >= std::min(s->tune.rel_min_records, s->n_inner4))) return w;
  // (the primary stage's rays are coherent: the top of the tree is in the caches whether staged or not, and a workgroup more
  // per CU is worth more than staged records -- C3 +2.8 % at 6 per CU with 6 records against 5 with 51)
  if (rel && w > 0) return s->tune.trace_wps;
  // deep trees (20 and more stack rows: C5, C3) leave a workgroup almost no LDS for the top of the tree at 6 per CU; one
  // workgroup less per CU stages 50 records instead of 10 (C5 +5 %, C3 +1 %; C2 and C4, 16 rows, lose 5 % at 5 per CU)
  int v = s->tune.trace_wps;
  while (v > 4 && records_staged4(s, v) < std::min(s->tune.min_staged, s->n_inner4)) v--;
  return v;
}
// rel: the launch traverses boxes translated by a common origin (traceq4_kernel<.., true>)
TraceCfg trace_cfg4(const EzrtScene* s, bool rel) {
  TraceCfg c;
  Tuning tu = s->tune;
  tu.trace_wps = wps4(s, rel);
  c.lds = (size_t)stack_rows4(s) * BLOCK * sizeof(int);
  const size_t lds_fixed = c.lds + BLOCK * sizeof(int);
  int blocks_per_cu = tu.trace_wps > 0 ? tu.trace_wps : 5;
  if ((size_t)blocks_per_cu * lds_fixed > 158 * 1024) blocks_per_cu = (int)((158 * 1024) / lds_fixed);
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  size_t lds_budget = (size_t)(158 * 1024) / blocks_per_cu;
  if (lds_budget > 64 * 1024) lds_budget = 64 * 1024;
  lds_budget -= lds_budget / 16;
  const size_t rec_bytes = (size_t)N4_LDS_DWORDS * 4;
  int n = lds_budget > lds_fixed ? (int)((lds_budget - lds_fixed) / rec_bytes) : 0;
  if (n > s->n_inner4) n = s->n_inner4;
  if (n > tu.lds_nodes) n = tu.lds_nodes;
  if (n < 0) n = 0;
  c.lds_nodes = n;
  c.lds_t = lds_fixed + (size_t)n * rec_bytes;
  c.blocks_per_cu = blocks_per_cu;
  c.grid_full = (unsigned)(s->num_cus * blocks_per_cu);
  return c;
}
// whether the timed stages of this scene run traceq4_kernel
bool use_wide4(const EzrtScene* s) {
  return s->tune.wide4 && s->n_inner4 > 0 && s->instr == 0 &&
         ((size_t)s->stack_need4 + 4) * BLOCK * sizeof(int) <= 60 * 1024; // stack rows (+ 3 of slack: prune 2) + lane table
}
// The instances of traceq4_kernel the library ships (round 6: 19, down from 72 -- the register budgets nobody ran, the cross-wave
// stealing variants and the scattered draw of two-ray queues are gone):
//   primary stage, rays generated in the launch (REL + GEN): 7 waves per SIMD with the default schedule (prune 2), 6 otherwise
//   a common origin without generation (gen_primary = 0, audit_via_queue = 2), bounce stages plain / SEMI: 6 waves, prune 0 / 1 / 2
//   bounce stages drawn in the scattered order (GS): the default schedule only
//   LOG (debug_stages = 2): the default schedule's five kernels
template <bool REL, bool GEN, bool SEMI, bool GS>
void launch_traceq4_p(int prune, bool log, int wps, dim3 grid, size_t lds, hipStream_t st, const TraceQ4Args& q) {
  const dim3 block(BLOCK);
  constexpr bool HAS_LOG = GEN || !REL; // (the default schedule's kernels)
  if (prune == 2 || GS) {
    if constexpr (REL && GEN) {
      if (wps >= 7) {
        if (log) hipLaunchKernelGGL((traceq4_kernel<7, REL, true, 2, GEN, SEMI, GS>), grid, block, lds, st, q);
        else hipLaunchKernelGGL((traceq4_kernel<7, REL, false, 2, GEN, SEMI, GS>), grid, block, lds, st, q);
        return;
      }
    }
    if constexpr (HAS_LOG) {
      if (log) {
        hipLaunchKernelGGL((traceq4_kernel<6, REL, true, 2, GEN, SEMI, GS>), grid, block, lds, st, q);
        return;
      }
    }
    hipLaunchKernelGGL((traceq4_kernel<6, REL, false, 2, GEN, SEMI, GS>), grid, block, lds, st, q);
    return;
  }
  if constexpr (!GS) {
    if (prune == 1) hipLaunchKernelGGL((traceq4_kernel<6, REL, false, 1, GEN, SEMI, false>), grid, block, lds, st, q);
    else hipLaunchKernelGGL((traceq4_kernel<6, REL, false, 0, GEN, SEMI, false>), grid, block, lds, st, q);
  }
}
template <bool REL, bool GEN>
void launch_traceq4_rel(EzrtScene* s, const TraceCfg& c, const TraceQ4Args& q, hipStream_t st) {
  const int wps = wps4(s, REL);
  const dim3 grid(c.grid_full);
  int prune = prune_mode(s);
  // (knob prune_mis: another order for the launches whose queue holds env shadow rays -- measured, not better)
  if (prune == 2 && q.q.rays_per_path == 2u && s->tune.prune_mis != 2) prune = s->tune.prune_mis;
  s->n_trace_launches++;
  const bool log = q.q.wave_log != nullptr; // (debug_stages=2)
  // rays with an exactly-zero direction component stay in this kernel (SEMI) where they come in numbers: the env shadow
  // rays of the MIS integrators' bounce stages (two rays per path); knob semi: 0 never, 2 every launch without a common origin
  const bool semi = !REL && !GEN && (s->tune.semi == 2 || (s->tune.semi == 1 && q.q.rays_per_path == 2u));
  // the scattered draw exists for the default schedule of the bounce stages: pruning with the nearest-first order, one ray per path
  const bool gs = !REL && !GEN && !semi && prune == 2 && q.gscat_shift != 0u;
  if (REL) launch_traceq4_p<REL, GEN, false, false>(prune, log, wps, grid, c.lds_t, st, q);
  else if (gs) launch_traceq4_p<false, false, false, true>(prune, log, wps, grid, c.lds_t, st, q);
  else if (semi) launch_traceq4_p<false, false, true, false>(prune, log, wps, grid, c.lds_t, st, q);
  else launch_traceq4_p<false, false, false, false>(prune, log, wps, grid, c.lds_t, st, q);
}
constexpr size_t QHEAD_SLOT_WORDS = (size_t)TRACE_HEADS * TRACE_HEAD_STRIDE;
constexpr size_t QHEADS_WORDS = 81 * QHEAD_SLOT_WORDS; // launch slots of reservation counters: stage b, redo launch 40 + b

// t: the stage's queue arguments as for the binary kernel (knobs already filled); rel: 4-wide records translated by
// t.origin (or NULL)
// gen (or NULL): the chunk's stage-0 arguments when the launch generates its primary rays itself (needs rel)
void fill_traceq4_args(const EzrtScene* s, const TraceCfg& c4, const TraceQArgs& t, const float4* rel, const WfArgs* gen, TraceQ4Args& A) {
  memset(&A.gen_p, 0, sizeof A.gen_p);
  A.gen_blocks = nullptr;
  A.gen_div_blocks = A.gen_div_sub = make_fastdiv(1u);
  A.gen_scatter = 1u;
  A.gen_scatter_shift = 6u;
  A.gen_frame_first = 0u;
  if (gen) {
    A.gen_p = gen->p;
    A.gen_blocks = gen->blocks;
    A.gen_div_blocks = gen->div_blocks;
    A.gen_div_sub = gen->div_sub;
    A.gen_scatter = gen->scatter;
    A.gen_scatter_shift = gen->scatter_shift;
    A.gen_frame_first = gen->frame_first;
  }
  A.q = t;
  A.q.stack_entries = (int32_t)(c4.lds / (BLOCK * sizeof(int)));
  A.q.lds_nodes = 0;
  A.q.inner_rel = nullptr;
  if (rel && s->tune.refill_min_rel > 0) A.q.refill_min = (uint32_t)s->tune.refill_min_rel;
  A.inner4 = s->inner4.p;
  A.inner4_rel = rel;
  A.root4 = s->root4;
  A.lds_nodes4 = c4.lds_nodes;
  {
    const double eps = 1.0 / 16777216.0;
    A.prune_cs = __builtin_nextafterf((float)(2.0 * 17.0 * eps), __builtin_inff());
    A.prune_a = s->prune_a;
    A.tri_leaf = (s->tune.tie_lca && s->tri_leaf.p && s->ref_up.p) ? s->tri_leaf.p : nullptr;
    A.ref_up = s->ref_up.p;
    A.stack_cap = (s->tune.debug_stack_cap > 0 && s->tune.debug_stack_cap < stack_cap4(s)) ? s->tune.debug_stack_cap : stack_cap4(s);
  }
  // (the even / odd slots of a two-ray path must stay in one granule: any granule >= 2 slots does)
  // knob bounce_scatter: 1 (default) = queues with one ray per path only.  Measured in the pipeline (profiles/r4/bounce_scatter_ab.txt):
  // C2 +2.7 % (trace launches 1.39 -> 1.345 ms), C3 -0.5 % (noise); the MIS integrators' queues (two rays per path sharing an
  // origin, env shadow rays that are coherent by construction) LOST 1.3 % (C4) and 2.8 % (C5) with it: never scattered
  A.gscat_shift = (!rel && !gen && !t.slot_map && s->tune.bounce_scatter != 0 && t.rays_per_path == 1u) ? 3u : 0u;
  A.handover = (s->tune.handover && t.steal) ? 1u : 0u;
  A.steal_bound = s->tune.steal_bound ? 1u : 0u;
}
void launch_traceq4_cfg(EzrtScene* s, const TraceCfg& c4, const TraceQArgs& t, const float4* rel, hipStream_t st, const WfArgs* gen = nullptr) {
  TraceQ4Args A;
  fill_traceq4_args(s, c4, t, rel, gen, A);
  if (rel && gen) launch_traceq4_rel<true, true>(s, c4, A, st);
  else if (rel) launch_traceq4_rel<true, false>(s, c4, A, st);
  else launch_traceq4_rel<false, false>(s, c4, A, st);
}

// schedule fields of a traceq launch that come from the knobs (clamped: ADVICE r1)
void fill_trace_knobs(const EzrtScene* s, const TraceCfg& c, TraceQArgs& t) {
  const Tuning& tu = s->tune;
  t.leaf_threshold = tu.leaf_threshold < 1 ? 1 : (tu.leaf_threshold > 64 ? 64 : tu.leaf_threshold);
  const int spct = s->chunk_pipelined ? tu.static_pct_pipelined : tu.static_pct;
  t.static_pct = (uint32_t)(spct < 0 ? 0 : (spct > 95 ? 95 : spct));
  t.refill_min = (uint32_t)(tu.refill_min < 1 ? 1 : (tu.refill_min > 64 ? 64 : tu.refill_min));
  t.pool_div = (uint32_t)(tu.pool_div < 1 ? 1 : tu.pool_div);
  t.pool_max = (uint32_t)(tu.pool_max < (int)TRACE_POOL_MIN ? (int)TRACE_POOL_MIN : (tu.pool_max > 4096 ? 4096 : tu.pool_max));
  t.pool_min = (uint32_t)(tu.pool_min < (int)TRACE_POOL_MIN ? (int)TRACE_POOL_MIN : (tu.pool_min > (int)t.pool_max ? (int)t.pool_max : tu.pool_min));
  t.stack_entries = (int32_t)(c.lds / (BLOCK * sizeof(int)));
  t.lds_nodes = c.lds_nodes;
  // distance pruning of the binary kernel's in-order traversal (redo launches, wide4 = 0): same margin as traceq4_kernel's
  t.anyhit_even = 0u; // (set by the render pipeline for the MIS integrators' bounce stages)
  t.prune_on = (s->prunable && tu.prune != 0 && s->instr == 0) ? 1u : 0u;
  t.prune_a = s->prune_a;
  t.prune_cs = __builtin_nextafterf((float)(2.0 * 17.0 / 16777216.0), __builtin_inff());
}

// ---- wavefront pipeline for one chunk of frames (all launches asynchronous on `st`)
// Shading kernels the library ships (round 6: 40 instances, down from 90).  The timed route: the primary stage and bounce 1 --
// the two big stages -- run shade_miss_kernel + shade_hit_kernel (leaving paths in 45 VGPRs, surface interactions in dense waves),
// the small later stages the fused shade_kernel (one launch less each).  The instrumented route (FULLCTR: the env-lookup
// counters of SURVEY 8(d)) runs the fused kernel in every stage.  (The fused kernel for the big stages, +12 % time, and the
// split pair for the small ones, -0.6 %, were knob `split_shade` until round 6.)
template <int INTEG>
void launch_shade_fused_i(const WfArgs& a, bool full, dim3 grid, hipStream_t st) {
  if (full) {
    if (a.bounce == 0) hipLaunchKernelGGL((shade_kernel<INTEG, true, 0>), grid, dim3(SHADE_BLOCK), 0, st, a);
    else if (a.bounce == 1) hipLaunchKernelGGL((shade_kernel<INTEG, true, 1>), grid, dim3(SHADE_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((shade_kernel<INTEG, true, 2>), grid, dim3(SHADE_BLOCK), 0, st, a);
  } else {
    hipLaunchKernelGGL((shade_kernel<INTEG, false, 2>), grid, dim3(SHADE_BLOCK), 0, st, a); // (bounce >= 2: see launch_shade)
  }
}
// `between` (or NULL): an event the second pass waits for -- the stage's redo launch on the side stream
// Returns the status of the cross-stream wait: if it failed, the second pass was NOT launched (it would read hit records
// the redo launch is still writing) and the caller fails the render call.
template <int INTEG, int STAGE>
hipError_t launch_shade_split_ib(const WfArgs& a, dim3 grid, dim3 grid_hit, hipStream_t st, hipEvent_t between) {
  hipLaunchKernelGGL((shade_miss_kernel<INTEG, false, STAGE>), grid, dim3(SHADE_BLOCK), 0, st, a);
  if (between) {
    const hipError_t e = hipStreamWaitEvent(st, between, 0);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((shade_hit_kernel<INTEG, false, STAGE>), grid_hit, dim3(SHADE_BLOCK), 0, st, a);
  return hipSuccess;
}
template <int INTEG>
hipError_t launch_shade_i(const WfArgs& a, bool full, dim3 grid, hipStream_t st, hipEvent_t between) {
  if (!full && a.bounce == 0) return launch_shade_split_ib<INTEG, 0>(a, grid, grid, st, between);
  if (!full && a.bounce == 1) return launch_shade_split_ib<INTEG, 1>(a, grid, grid, st, between);
  if (between) { // (the fused kernel reads every hit record at once: behind the redo launch)
    const hipError_t e = hipStreamWaitEvent(st, between, 0);
    if (e != hipSuccess) return e;
  }
  launch_shade_fused_i<INTEG>(a, full, grid, st);
  return hipSuccess;
}
// whether stage b's shading is the split pair (the caller's redo launch may then overlap the first pass)
inline bool shade_is_split(bool full, int bounce) { return !full && bounce <= 1; }
hipError_t launch_shade(const WfArgs& a, bool full, dim3 grid, hipStream_t st, hipEvent_t between) {
  switch (a.p.integrator) {
    case EZRT_INTEGRATOR_P3_DIFFUSE: return launch_shade_i<EZRT_INTEGRATOR_P3_DIFFUSE>(a, full, grid, st, between);
    case EZRT_INTEGRATOR_P4_DISNEY: return launch_shade_i<EZRT_INTEGRATOR_P4_DISNEY>(a, full, grid, st, between);
    case EZRT_INTEGRATOR_P5_SOBOL: return launch_shade_i<EZRT_INTEGRATOR_P5_SOBOL>(a, full, grid, st, between);
    case EZRT_INTEGRATOR_P5_MIS_ANISO: return launch_shade_i<EZRT_INTEGRATOR_P5_MIS_ANISO>(a, full, grid, st, between);
    default: return launch_shade_i<EZRT_INTEGRATOR_P5_MIS>(a, full, grid, st, between);
  }
}

// The queues of one chunk (n_slots pixel-samples in flight).  hipErrorOutOfMemory leaves the pipe consistent (a DevBuf that
// failed to grow is empty), so the caller can retry with a smaller chunk.
hipError_t ensure_chunk_scratch(EzrtScene* s, Pipe& pp, size_t n_slots, bool mis, hipStream_t st) {
  if (s->tune.debug_oom_above > 0 && n_slots > (size_t)s->tune.debug_oom_above) return hipErrorOutOfMemory; // (test hook)
  const size_t n_rays_max = n_slots * (mis ? 2 : 1);
#define EZ_ENSURE(x)                   \
  do {                                 \
    hipError_t e_ = (x);               \
    if (e_ != hipSuccess) return e_;   \
  } while (0)
  EZ_ENSURE(pp.samples.ensure(n_slots));
  for (int k = 0; k < 2; k++) {
    EZ_ENSURE(pp.rq_o[k].ensure(n_slots)); // (one origin per path)
    EZ_ENSURE(pp.rq_d[k].ensure(n_rays_max));
    for (int j = 0; j < (mis ? 5 : 4); j++) EZ_ENSURE(pp.st[k][j].ensure(n_slots));
  }
  EZ_ENSURE(pp.hits2[0].ensure(n_rays_max));
  EZ_ENSURE(pp.hits2[1].ensure(n_rays_max));
  EZ_ENSURE(pp.redo_slots.ensure(n_rays_max));
  if (pp.redo_flag.n < n_rays_max) { // zeroed once; every entry set is cleared again by the redo launch
    EZ_ENSURE(pp.redo_flag.ensure(n_rays_max));
    EZ_ENSURE(hipMemsetAsync(pp.redo_flag.p, 0, n_rays_max * sizeof(uint32_t), st));
    EZ_ENSURE(hipStreamSynchronize(st)); // (the chunk may run on another stream than `st`: pipeline_calls; growth is rare)
  }
  EZ_ENSURE(pp.defer_list.ensure(n_slots + (size_t)2048 * 1024)); // per-workgroup regions: iterations x SHADE_BLOCK each
  EZ_ENSURE(pp.defer_count.ensure(2048u * 1024u / SHADE_BLOCK));
#undef EZ_ENSURE
  return hipSuccess;
}

void release_chunk_scratch(Pipe& pp) {
  pp.samples.release();
  for (int k = 0; k < 2; k++) {
    pp.rq_o[k].release();
    pp.rq_d[k].release();
    for (auto& b : pp.st[k]) b.release();
    pp.hits2[k].release();
  }
  pp.redo_slots.release();
  pp.redo_flag.release();
  pp.defer_list.release();
  pp.defer_count.release();
}

struct PathLogTarget { // ezrt_render_paths through the timed pipeline (audit_via_queue)
  int32_t* tri = nullptr;
  float* t = nullptr;
  float* colour = nullptr;
};
int wavefront_chunk(EzrtScene* s, Pipe& pp, const EzrtRenderParams* p, int nb, uint32_t frame_first, uint32_t nf, hipStream_t st,
                    const PathLogTarget* plog = nullptr) {
  const bool mis = p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO;
  const bool full = s->instr > 0;
  const size_t n_slots = (size_t)nb * BLOCK * nf;
  if (p->max_bounce > 32) return fail(EZRT_ERR_UNSUPPORTED, "max_bounce > 32: more stages than this build has queue counters for");
  HIP_TRY(ensure_chunk_scratch(s, pp, n_slots, mis, st)); // (a no-op after ezrt_render_device's sizing pass)
  // [0..63] paths per stage, [64..99] queue heads, [100..119] debug,
  // [128..] redo counts per stage, [192..] redo queue heads per stage
  constexpr size_t HEAD_SLOT = (size_t)TRACE_HEADS * TRACE_HEAD_STRIDE; // launch slots: stage b, redo 40 + b
  HIP_TRY(pp.qheads.ensure(QHEADS_WORDS)); // (both zeroed by raygen_kernel: ChunkPrologue)
  HIP_TRY(pp.qcounts.ensure(320));
  {
    int rc_cu = ensure_num_cus(s);
    if (rc_cu) return rc_cu;
  }
  auto queue = [&](int k) {
    RayQueue q;
    q.o = pp.rq_o[k].p;
    q.d = pp.rq_d[k].p;
    return q;
  };
  auto state = [&](int k) {
    PathState t;
    t.s0 = pp.st[k][0].p;
    t.s1 = pp.st[k][1].p;
    t.s2 = pp.st[k][2].p;
    t.s3 = pp.st[k][3].p;
    t.s4 = pp.st[k][4].p;
    return t;
  };
  WfArgs a;
  a.sc = s->dev();
  a.p = *p;
  a.blocks = s->blocks.p;
  a.n_blocks = nb;
  a.frame_first = frame_first;
  a.n_slots = (uint32_t)n_slots;
  a.samples = pp.samples.p;
  a.counters = s->counters.p;
  a.hits = pp.hits2[1].p;
  a.hits_out = reinterpret_cast<unsigned long long*>(pp.hits2[0].p);
  // raygen -> queue 0
  a.rq_in = queue(1);
  a.rq_out = queue(0);
  a.st_in = state(1);
  a.st_out = state(0);
  a.n_in = pp.qcounts.p;
  a.n_out = pp.qcounts.p;
  a.bounce = 0;
  a.scatter = 1u;
  a.scatter_shift = 6u;
  if (s->tune.scatter) { // multiplier coprime to n_sub
    a.scatter_shift = (uint32_t)(s->tune.scatter >= 4 && s->tune.scatter <= 8 ? s->tune.scatter : 6); // 1: 8x8 sub-blocks
    const uint32_t n_sub = ((uint32_t)nb * 256u) >> a.scatter_shift;
    auto gcd = [](uint32_t x, uint32_t y) {
      while (y) {
        const uint32_t t = x % y;
        x = y;
        y = t;
      }
      return x;
    };
    uint32_t m = 2531u; // prime; consecutive queue granules land 2531 sub-blocks apart.  r * m must fit 32 bits:
    if (n_sub > (1u << 20)) m = 1u; // (frames beyond 2^20 sub-blocks = 8k x 8k pixels keep the raster order)
    while (m > 1u && gcd(m, n_sub) != 1u) m += 2u;
    a.scatter = m % n_sub ? m % n_sub : 1u;
  }
  a.div_blocks = make_fastdiv((uint32_t)nb);
  a.div_sub = make_fastdiv(((uint32_t)nb * 256u) >> a.scatter_shift);
  HIP_TRY(pp.sobol_tab.ensure((size_t)nf * 16));
  a.sobol_tab = pp.sobol_tab.p;
  a.sobol_out = pp.sobol_tab.p;
  a.n_frames = nf;
  const bool wide = use_wide4(s);
  if (!wide && s->tune.rel_boxes && s->n_inner > 0) {
    HIP_TRY(pp.inner_rel.ensure((size_t)s->n_inner * 4));
    hipLaunchKernelGGL(inner_rel_kernel, dim3((unsigned)((s->n_inner + 255) / 256)), dim3(256), 0, st, s->inner.p, s->n_inner,
                       p->eye[0], p->eye[1], p->eye[2], pp.inner_rel.p);
  }
  const TraceCfg cfg4_rel = wide ? trace_cfg4(s, true) : TraceCfg(), cfg4_abs = wide ? trace_cfg4(s, false) : TraceCfg();
  ChunkPrologue pro;
  pro.zero_a = pp.qheads.p;
  pro.n_zero_a = (uint32_t)QHEADS_WORDS;
  pro.zero_b = pp.qcounts.p;
  pro.n_zero_b = 320u;
  pro.inner4 = nullptr;
  pro.inner4_rel = nullptr;
  pro.n_inner4 = 0;
  pro.sx = p->eye[0];
  pro.sy = p->eye[1];
  pro.sz = p->eye[2];
  if (wide && s->tune.rel_boxes) {
    HIP_TRY(pp.inner4_rel.ensure((size_t)s->n_inner4 * N4_FLOAT4));
    pro.inner4 = s->inner4.p;
    pro.inner4_rel = pp.inner4_rel.p;
    pro.n_inner4 = s->n_inner4;
  }
  // primary rays generated where they are consumed (primary_dir) when stage 0 runs the 4-wide kernel on eye-relative records
  const bool gen_primary = wide && s->tune.rel_boxes && s->tune.gen_primary;
  a.all_owned = (p->shard_count <= 1 && p->x0 == 0 && p->y0 == 0 && p->x1 == p->width && p->y1 == p->height && p->width % 16 == 0 &&
                 p->height % 16 == 0 && s->tune.lazy_dir) ? 1u : 0u;
  a.gen_primary = 0u; // (the shading passes read the directions the trace launch stored: see traceq4_kernel GEN)
  if (gen_primary) hipLaunchKernelGGL(chunk_prologue_kernel, dim3((unsigned)(4 * s->num_cus)), dim3(BLOCK), 0, st, a, pro);
  else hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((n_slots + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, a, pro);
  if (s->tune.debug_stages && wide)
    fprintf(stderr, "[ezrt] traceq4 launches: %d stack rows (binary tree depth %d); primary stage %d workgroups/CU, %zu B LDS, %d records staged; "
            "bounce stages %d workgroups/CU, %zu B LDS, %d records staged\n", s->stack_need4, s->depth, cfg4_rel.blocks_per_cu, cfg4_rel.lds_t,
            cfg4_rel.lds_nodes, cfg4_abs.blocks_per_cu, cfg4_abs.lds_t, cfg4_abs.lds_nodes);

  const Tuning& tu = s->tune;
  const TraceCfg cfg = trace_cfg(s);
  const int debug_stages = tu.debug_stages;
  const unsigned trace_grid_full = cfg.grid_full;
  unsigned shade_grid = (unsigned)((n_slots + SHADE_BLOCK - 1) / SHADE_BLOCK);
  unsigned shade_grid_max = s->tune.shade_wgs > 0 ? (unsigned)s->tune.shade_wgs : (unsigned)(12 * s->num_cus);
  if (shade_grid_max > 4096u) shade_grid_max = 4096u; // (the defer lists are sized for that)
  if (shade_grid_max < 1u) shade_grid_max = 1u;
  if (shade_grid > shade_grid_max) shade_grid = shade_grid_max;

  for (int b = 0; b <= p->max_bounce; b++) {
    const int in = b & 1, out = in ^ 1;
    TraceQArgs t;
    t.sc = trace_scene(a.sc);
    t.rq = queue(in);
    t.hits = pp.hits2[in].p;
    t.n_paths = pp.qcounts.p + b;
    t.rays_per_path = (mis && b > 0) ? 2u : 1u;
    t.const_origin = b == 0 ? 1u : (mis ? 2u : 0u); // (MIS: one stored origin per path, shared by its two rays)
    t.inner_rel = (!wide && b == 0 && tu.rel_boxes && s->n_inner > 0) ? pp.inner_rel.p : nullptr;
    t.origin[0] = p->eye[0];
    t.origin[1] = p->eye[1];
    t.origin[2] = p->eye[2];
    t.head = pp.qheads.p + (size_t)b * HEAD_SLOT;
    t.counters = s->counters.p;
    fill_trace_knobs(s, cfg, t);
    t.anyhit_even = (mis && b > 0 && wide && !plog && !full && tu.anyhit) ? 1u : 0u;
    t.dbg = debug_stages ? (pp.qcounts.p + 100 + 4 * (b & 3)) : nullptr;
    t.slot_map = nullptr;
    t.steal = tu.steal ? 1u : 0u;
    t.count_rays = 1u;
    t.redo_count = pp.qcounts.p + 128 + b;
    t.redo_slots = pp.redo_slots.p;
    t.redo_flag = pp.redo_flag.p;
    t.force_pending = (uint32_t)tu.debug_force_pending;
    t.wave_log = nullptr;
    if (debug_stages >= 2) {
      HIP_TRY(pp.wave_log.ensure((size_t)trace_grid_full * (BLOCK / 64) * 8));
      HIP_TRY(hipMemsetAsync(pp.wave_log.p, 0, (size_t)trace_grid_full * (BLOCK / 64) * 8 * sizeof(unsigned long long), st));
      t.wave_log = pp.wave_log.p;
    }
    auto launch_traceq = [&](const TraceQArgs& q, bool small = false) { launch_traceq_cfg(s, cfg, q, small, st); };
    const bool split_here = shade_is_split(full, b);
    // the redo launch under the first shading pass: only where the first pass cannot be misled by a record the redo
    // launch is still to write -- traceq4_kernel marks those HIT_PENDING -- and only in plain timed runs
    const bool overlap_redo = wide && split_here && tu.redo_overlap && !full && !plog && !debug_stages;
    hipEvent_t ev_between = nullptr;
    int e = tu.launch_events ? s->n_trace_events : MAX_TRACE_EVENTS;
    if (e < MAX_TRACE_EVENTS) {
      while (s->n_trace_events_created <= e) { // (calls with more than 64 timed launches: created on first use)
        const int ne = s->n_trace_events_created;
        if (!s->ev_trace[ne][0]) HIP_TRY(hipEventCreate(&s->ev_trace[ne][0]));
        if (!s->ev_trace[ne][1]) HIP_TRY(hipEventCreate(&s->ev_trace[ne][1]));
        s->n_trace_events_created = ne + 1;
      }
      HIP_TRY(hipEventRecord(s->ev_trace[e][0], st));
    }
    {
      if (wide) {
        const bool rel = b == 0 && tu.rel_boxes;
        launch_traceq4_cfg(s, rel ? cfg4_rel : cfg4_abs, t, rel ? pp.inner4_rel.p : nullptr, st, (rel && gen_primary) ? &a : nullptr);
      }
      else launch_traceq(t);
      if (t.steal || wide) { // rays that met an exact distance tie, or (4-wide) are not tame -- normally none: reference order, plain stores
        TraceQArgs r = t;
        r.steal = 0u;
        r.count_rays = 0u;
        r.slot_map = pp.redo_slots.p;
        r.n_paths = pp.qcounts.p + 128 + b;
        r.rays_per_path = 1u;
        r.head = pp.qheads.p + (size_t)(40 + b) * HEAD_SLOT;
        r.dbg = nullptr;
        r.wave_log = nullptr;
        r.force_pending = 0u;
        if (overlap_redo) {
          // the redo launches are a handful of rays on the critical path of the stage's second shading pass: highest priority
          if (!pp.side) HIP_TRY(ezh::stream_acquire(true, &pp.side, &pp.stream_device));
          HIP_TRY(hipEventRecord(pp.ev_main, st));
          HIP_TRY(hipStreamWaitEvent(pp.side, pp.ev_main, 0));
          launch_traceq_cfg(s, cfg, r, true, pp.side);
          HIP_TRY(hipEventRecord(pp.ev_redo, pp.side));
          ev_between = pp.ev_redo;
        } else {
          launch_traceq(r, true);
        }
      }
    }
    if (e < MAX_TRACE_EVENTS) {
      HIP_TRY(hipEventRecord(s->ev_trace[e][1], st));
      s->n_trace_events++;
    }
    if (plog) { // audit: this stage's hit records, exactly as the trace (+ redo) launches left them
      PathLogArgs g;
      g.hits = pp.hits2[in].p;
      g.rq_d = queue(in).d;
      const bool compact = p->integrator == EZRT_INTEGRATOR_P5_SOBOL; // (compact_state<50>: see PathState)
      const bool mis1 = mis && b == 1; // (mis_stage1_state: the sample slot is s1.y)
      g.st_slot = (compact || mis1) ? state(in).s1 : state(in).s2;
      g.slot_comp = compact ? (b == 1 ? 0 : 3) : (mis1 ? 1 : 3);
      g.slot_stride = (compact && b == 1) ? 2 : 4;
      g.n_in = pp.qcounts.p + b;
      g.n_slots = (uint32_t)n_slots;
      g.bounce = b;
      g.mis = mis ? 1 : 0;
      g.blocks = s->blocks.p;
      g.n_blocks = nb;
      g.frame_first = frame_first;
      g.scatter = a.scatter;
      g.scatter_shift = a.scatter_shift;
      g.div_blocks = a.div_blocks;
      g.div_sub = a.div_sub;
      g.width = p->width;
      g.p = *p;
      g.log_slots = 1 + 2 * p->max_bounce;
      g.log_tri = plog->tri;
      g.log_t = plog->t;
      g.log_colour = plog->colour;
      g.samples = pp.samples.p;
      hipLaunchKernelGGL(pathlog_kernel, dim3((unsigned)((n_slots + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, g);
    }
    a.hits = pp.hits2[in].p;
    a.hits_out = reinterpret_cast<unsigned long long*>(pp.hits2[out].p);
    a.rq_in = queue(in);
    a.rq_out = queue(out);
    a.st_in = state(in);
    a.st_out = state(out);
    a.n_in = pp.qcounts.p + b;
    a.n_out = pp.qcounts.p + b + 1;
    a.bounce = b;
    a.defer_list = pp.defer_list.p;
    a.defer_count = pp.defer_count.p;
    HIP_TRY(launch_shade(a, full, dim3(shade_grid), st, ev_between));
    if (debug_stages) { // diagnostic only: per-stage queue sizes and counters (synchronises)
      uint32_t q[2] = {0, 0};
      unsigned long long c[EZRT_CTR_COUNT];
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipMemcpy(q, pp.qcounts.p + b, sizeof q, hipMemcpyDeviceToHost));
      {
        unsigned long long all[CTR_SLOTS * EZRT_CTR_COUNT];
        HIP_TRY(hipMemcpy(all, s->counters.p, sizeof all, hipMemcpyDeviceToHost));
        for (int k = 0; k < EZRT_CTR_COUNT; k++) {
          c[k] = 0;
          for (int j = 0; j < CTR_SLOTS; j++) c[k] += all[j * EZRT_CTR_COUNT + k];
        }
      }
      if (debug_stages >= 2 && t.wave_log) { // per-wave life times of this stage's traceq launch
        const size_t nw = (size_t)trace_grid_full * (BLOCK / 64);
        std::vector<unsigned long long> w(nw * 8);
        HIP_TRY(hipMemcpy(w.data(), pp.wave_log.p, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < nw; i++)
          if (w[i * 8] && w[i * 8] < t0) t0 = w[i * 8];
        unsigned long long s_it = 0, s_is = 0, s_il = 0, s_ll = 0, s_lr = 0, s_busy = 0, s_rf = 0, s_st = 0;
        std::vector<double> endt, life, its, rays, startt, exht, after;
        for (size_t i = 0; i < nw; i++)
          if (w[i * 8]) {
            startt.push_back((double)(w[i * 8] - t0) * 0.01);
            endt.push_back((double)(w[i * 8 + 1] - t0) * 0.01);
            life.push_back((double)(w[i * 8 + 1] - w[i * 8]) * 0.01);
            if (w[i * 8 + 7]) {
              exht.push_back((double)(w[i * 8 + 7] - t0) * 0.01);
              after.push_back((double)(w[i * 8 + 1] - w[i * 8 + 7]) * 0.01);
            }
            its.push_back((double)(uint32_t)w[i * 8 + 2]);
            rays.push_back((double)(uint32_t)w[i * 8 + 3]);
            s_it += (uint32_t)w[i * 8 + 2];
            s_is += w[i * 8 + 2] >> 32;
            s_il += w[i * 8 + 3] >> 32;
            s_ll += (uint32_t)w[i * 8 + 4];
            s_lr += w[i * 8 + 4] >> 32;
            s_busy += (uint32_t)w[i * 8 + 5];
            s_rf += (uint32_t)w[i * 8 + 6];
            s_st += w[i * 8 + 6] >> 32;
          }
        auto pct = [](std::vector<double>& v, double q) {
          if (v.empty()) return 0.0;
          std::sort(v.begin(), v.end());
          return v[(size_t)(q * (double)(v.size() - 1))];
        };
        fprintf(stderr, "[ezrt]   waves %zu | start us p50 %.1f max %.1f | end us p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | life us p50 %.1f max %.1f | iters p50 %.0f p99 %.0f max %.0f | rays p50 %.0f max %.0f\n",
                endt.size(), pct(startt, 0.5), pct(startt, 1.0), pct(endt, 0.1), pct(endt, 0.5), pct(endt, 0.9), pct(endt, 0.99),
                pct(endt, 1.0), pct(life, 0.5), pct(life, 1.0), pct(its, 0.5), pct(its, 0.99), pct(its, 1.0), pct(rays, 0.5),
                pct(rays, 1.0));
        fprintf(stderr, "[ezrt]   iterations %llu: lanes with a ray %.1f/64 | inner steps in %.0f %% of them, %.1f lanes each | cooperative leaf rounds in %.0f %%, %.1f rays each\n",
                s_it, (double)s_busy / (double)(s_it ? s_it : 1), 100.0 * (double)s_is / (double)(s_it ? s_it : 1), (double)s_il / (double)(s_is ? s_is : 1),
                100.0 * (double)s_lr / (double)(s_it ? s_it : 1), (double)s_ll / (double)(s_lr ? s_lr : 1));
        if (!exht.empty())
          fprintf(stderr, "[ezrt]   queue found empty at us p10 %.1f p50 %.1f p90 %.1f max %.1f | a wave then runs on for us p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
                  pct(exht, 0.1), pct(exht, 0.5), pct(exht, 0.9), pct(exht, 1.0), pct(after, 0.1), pct(after, 0.5), pct(after, 0.9), pct(after, 1.0));
        fprintf(stderr, "[ezrt]   refill block in %.0f %% of the iterations, steal block in %.0f %%\n",
                100.0 * (double)s_rf / (double)(s_it ? s_it : 1), 100.0 * (double)s_st / (double)(s_it ? s_it : 1));
      }
      uint32_t dbg[3] = {0, 0, 0};
      HIP_TRY(hipMemcpy(dbg, pp.qcounts.p + 100 + 4 * (b & 3), sizeof dbg, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemset(pp.qcounts.p + 100 + 4 * (b & 3), 0, sizeof dbg));
      {
        uint32_t redo_n = 0;
        HIP_TRY(hipMemcpy(&redo_n, pp.qcounts.p + 128 + b, sizeof redo_n, hipMemcpyDeviceToHost));
        fprintf(stderr, "[ezrt] stage %d: %u rays re-traced in reference order (exact ties / not tame)\n", b, redo_n);
      }
      fprintf(stderr, "[ezrt] stage %d: paths_in %u paths_out %u | cum rays %llu pops %llu inner %llu tris %llu | max/ray pops %u tris %u iters %u\n", b, q[0],
              q[1], c[0], c[1], c[2], c[3], dbg[0], dbg[1], dbg[2]);
    }
  }
  if (plog && plog->colour) {
    PathLogArgs g;
    memset(&g, 0, sizeof g);
    g.blocks = s->blocks.p;
    g.n_blocks = nb;
    g.div_blocks = a.div_blocks;
    g.div_sub = a.div_sub;
    g.frame_first = frame_first;
    g.width = p->width;
    g.log_colour = plog->colour;
    g.samples = pp.samples.p;
    hipLaunchKernelGGL(pathcolour_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, g, *p);
  }
  return 0;
}

} // namespace

extern "C" {

const char* ezrt_last_error(void) { return g_err; }
// for the other translation units of this library (ezrt_lbvh.hip)
__attribute__((visibility("hidden"))) int ezrt_fail_msg(int code, const char* msg) { return fail(code, "%s", msg); }
const char* ezrt_backend(void) { return "hip:gfx950"; }

static int scene_create_impl(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out);
int ezrt_scene_create(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out) {
  if (!out) return fail(EZRT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  try { // (host-side allocations and worker threads: no exception crosses the C ABI)
    return scene_create_impl(tri, n_tri, nodes, n_nodes, out);
  } catch (const std::bad_alloc&) {
    return fail(EZRT_ERR_NOMEM, "out of host memory while building the scene's records");
  } catch (const std::exception& e) {
    return fail(EZRT_ERR_DEVICE, "scene creation failed: %s", e.what());
  }
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

int ezrt_trim(void) { return ezh::stream_pool_trim(); }

void ezrt_scene_destroy(EzrtScene* s) {
  if (!s) return;
  // (whatever ensure_events got to create: it may have stopped half way)
  if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
  if (s->ev_end) (void)hipEventDestroy(s->ev_end);
  for (Pipe& q : s->pipe) {
    if (q.side) ezh::stream_park(q.side, true, q.stream_device); // (q.stream is the device's shared pair: released below, not parked)
    if (q.ev_main) (void)hipEventDestroy(q.ev_main);
    if (q.ev_redo) (void)hipEventDestroy(q.ev_redo);
    if (q.ev_done) (void)hipEventDestroy(q.ev_done);
    if (q.ev_free) (void)hipEventDestroy(q.ev_free);
  }
  for (int i = 0; i < MAX_TRACE_EVENTS; i++) {
    if (s->ev_trace[i][0]) (void)hipEventDestroy(s->ev_trace[i][0]);
    if (s->ev_trace[i][1]) (void)hipEventDestroy(s->ev_trace[i][1]);
  }
  if (s->pipe[0].stream) ezh::stream_shared_release(s->pipe[0].stream_device);
  delete s;
}

int ezrt_scene_set_env(EzrtScene* s, const float* hdr, const float* cache, int w, int h, int filter) {
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

int ezrt_render_device(EzrtScene* s, const EzrtRenderParams* p, float* accum_dev, void* stream) {
  if (!s || !accum_dev) return fail(EZRT_ERR_INVALID, "scene/accum is NULL");
  int rc = validate_params(s, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  rc = ensure_events(s);
  if (rc) return rc;
  rc = build_blocks(s, *p, st);
  if (rc) return rc;
  s->timed = false;
  s->n_trace_events = 0;
  s->n_trace_launches = 0;
  const int nb = (int)s->blocks_host.size();
  HIP_TRY(hipEventRecord(s->ev_begin, st));
  if (nb > 0 && p->spp > 0) {
    // frames per chunk: at most 2^chunk_log2 pixel-samples in flight (see Tuning)
    const size_t per_frame = (size_t)nb * BLOCK;
    size_t chunk = ((size_t)1 << s->tune.chunk_log2) / per_frame;
    if (chunk < 1) chunk = 1;
    if (chunk > p->spp) chunk = p->spp;
    const int use_mega = s->tune.megakernel;
    // Chunks pipelined ACROSS calls (knob pipeline_calls, round 4): chunk i of the scene's life runs on scratch set i & 1 and that
    // set's own stream.  Nothing it does touches the caller's memory -- it reads the scene and writes its own queues and samples --
    // so it need not wait for anything the caller queued before this call; the kernel that DOES touch the caller's frame buffer,
    // accumulate_kernel, stays on the caller's stream, after a wait for the chunk's samples, so the frame buffer sees the calls
    // in the order they were made and everything the caller queues behind a call finds it complete.  A scratch set is reused
    // only after the accumulation that read its samples (ev_free).  What it buys: the small late stages of a chunk last as long as
    // their deepest rays (section 6 of DESIGN.md) and leave most of the chip idle; the next chunk's primary stage now runs under
    // them.  Two independent scenes on two streams showed the potential first: C2 +10-11 % aggregate, C4 +-0 (tools/exp_two_streams.py).
    // Measured (profiles/r4/pipeline_calls_ab.txt): C2 12.7 -> 14.4-14.6 Grays/s (+13-14 %).  With the trace queues dealt half
    // statically (the unpipelined optimum) C3 / C5 / C4-at-256-spp LOST 2.4 / 1.0 / 5 %: a pipelined chunk's persistent trace
    // workgroups become resident only as the other chunk's launches free wave slots, and the pools dealt statically to a
    // workgroup that arrives late are the launch's tail.  With all-dynamic queues for pipelined chunks (static_pct_pipelined
    // = 0) every config gains: C3 +3.0 %, C4 +2.6 %, C5 +1.8 %, C2 unchanged at +13 %.  So every scene is pipelined.
    // (not with per-launch timing events: they sit on the chunk's stream while the call's begin / end events sit on the caller's, and two
    // overlapping chunks would have their launch intervals summed twice -- ezrt_last_render_ms describes calls run one chunk at a time)
    const bool xcall = !use_mega && !s->tune.debug_stages && s->tune.pipeline_calls != 0 && !s->tune.launch_events;
    // Two chunks in flight.  (Round 5 measured three and four -- knob pipeline_depth, removed in round 6: a BURST of three calls gained
    // 4 % with a third scratch set, the steady state of back-to-back calls was identical to four digits, the 1/2 .. 1/8 shards of a
    // frame LOST 3-10 %, and a third set costs up to 23 GB: profiles/r5/pipeline_depth_ab.txt.  So did splitting one call's frames into
    // sub-chunks on two streams -- knob pipes, round 2: every stage's latency-bound end is paid twice, 3.81 vs 3.65 ms.)
    constexpr int depth = 2;
    const int n_scratch = xcall ? depth : 1;
    if (!use_mega) { // size the chunk's queues now: if they do not fit, halve the chunk (same results, more launches)
      const bool mis = p->integrator == EZRT_INTEGRATOR_P5_MIS || p->integrator == EZRT_INTEGRATOR_P5_MIS_ANISO;
      for (;;) {
        hipError_t e = hipSuccess;
        for (int i = 0; i < n_scratch && e == hipSuccess; i++) e = ensure_chunk_scratch(s, s->pipe[i], per_frame * chunk, mis, st);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || chunk <= 1)
          return fail(EZRT_ERR_DEVICE, "render scratch for %zu pixel-samples in flight: %s", per_frame * chunk, hipGetErrorString(e));
        // DevBuf::ensure never shrinks: the buffers that did fit at the failed size would stay allocated and the smaller
        // request could fail where a clean allocation fits -- give everything back first
        HIP_TRY(hipStreamSynchronize(st));
        for (Pipe& q : s->pipe) release_chunk_scratch(q);
        chunk = (chunk + 1) / 2;
      }
    }
    const size_t lds = stack_lds_bytes(s);
    for (uint32_t done = 0; done < p->spp;) {
      uint32_t nf = (uint32_t)((p->spp - done < chunk) ? (p->spp - done) : chunk);
      Pipe& q = s->pipe[xcall ? (s->chunk_seq % (uint32_t)depth) : 0u];
      hipStream_t qs = xcall ? q.stream : st;
      HIP_TRY(q.samples.ensure(per_frame * chunk));
      // after the accumulation that consumed this scratch set's previous samples: two chunks ago when pipelined; in the plain route
      // (launch_events, debug_stages, knob off) the previous user may have been a pipelined call whose accumulation sits on ANOTHER
      // caller stream (toggling launch_events, as bench.py does, made that race easy to reach)
      if (q.free_recorded) HIP_TRY(hipStreamWaitEvent(qs, q.ev_free, 0));
      if (use_mega) {
        TraceArgs a;
        a.sc = s->dev();
        a.p = *p;
        a.blocks = s->blocks.p;
        a.n_blocks = nb;
        a.frame_first = p->frame0 + done;
        a.samples = q.samples.p;
        a.counters = s->counters.p;
        a.log_tri = nullptr;
        a.log_t = nullptr;
        a.log_colour = nullptr;
        a.stack_entries = s->depth;
        int e = s->n_trace_events;
        if (e < s->n_trace_events_created) HIP_TRY(hipEventRecord(s->ev_trace[e][0], st));
        launch_trace(a, s->instr > 0 ? 1 : 0, dim3((unsigned)((size_t)nb * nf)), lds, st);
        if (e < s->n_trace_events_created) {
          HIP_TRY(hipEventRecord(s->ev_trace[e][1], st));
          s->n_trace_events++;
        }
        s->n_trace_launches++;
      } else {
        s->chunk_pipelined = xcall;
        rc = wavefront_chunk(s, q, p, nb, p->frame0 + done, nf, qs);
        s->chunk_pipelined = false;
        if (rc) return rc;
      }
      if (xcall) { // the running mean is applied in frame order, on the caller's stream
        HIP_TRY(hipEventRecord(q.ev_done, qs));
        HIP_TRY(hipStreamWaitEvent(st, q.ev_done, 0));
      }
      AccumArgs b;
      b.p = *p;
      b.blocks = s->blocks.p;
      b.n_blocks = nb;
      b.frame_first = p->frame0 + done;
      b.n_frames = nf;
      b.samples = q.samples.p;
      b.accum = reinterpret_cast<float4*>(accum_dev);
      hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, b);
      // (recorded in every mode: a later pipelined chunk on this scratch set's own stream must wait for THIS use of it too)
      HIP_TRY(hipEventRecord(q.ev_free, st));
      q.free_recorded = true;
      s->chunk_seq++;
      done += nf;
    }
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(s->ev_end, st));
  s->timed = true;
  return 0;
}

int ezrt_render(EzrtScene* s, const EzrtRenderParams* p, float* accum) {
  if (!s || !accum) return fail(EZRT_ERR_INVALID, "scene/accum is NULL");
  int rc = validate_params(s, p);
  if (rc) return rc;
  size_t n = (size_t)p->width * p->height;
  HIP_TRY(s->accum_tmp.ensure(n));
  HIP_TRY(hipMemcpy(s->accum_tmp.p, accum, n * sizeof(float4), hipMemcpyHostToDevice));
  rc = ezrt_render_device(s, p, reinterpret_cast<float*>(s->accum_tmp.p), nullptr);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr)); // the call's stream (side-stream launches are joined to it by events), not the device
  HIP_TRY(hipMemcpy(accum, s->accum_tmp.p, n * sizeof(float4), hipMemcpyDeviceToHost));
  return 0;
}

int ezrt_frame_create(int width, int height, float** frame_dev) {
  if (!frame_dev || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  *frame_dev = nullptr;
  const size_t bytes = (size_t)width * height * sizeof(float4);
  float* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, bytes));
  hipError_t e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(EZRT_ERR_DEVICE, "hipMemset failed: %s", hipGetErrorString(e));
  }
  *frame_dev = p;
  return 0;
}
int ezrt_frame_destroy(float* frame_dev) {
  if (frame_dev) HIP_TRY(hipFree(frame_dev));
  return 0;
}
// Synchronise the device that OWNS the frame (whatever stream rendered into it), not whichever device happens to be
// current: the frame may have been created and rendered while another device was current (ezrt_mgpu, a host that
// switches devices).  The caller's current device is restored on every exit.
static int frame_copy(float* frame_dev, float* rgba_host, size_t bytes, bool to_host) {
  hipPointerAttribute_t at;
  HIP_TRY(hipPointerGetAttributes(&at, frame_dev));
  int prev = 0;
  HIP_TRY(hipGetDevice(&prev));
  struct Restore {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  if (at.device != prev) HIP_TRY(hipSetDevice(at.device));
  HIP_TRY(hipDeviceSynchronize());
  if (to_host) HIP_TRY(hipMemcpy(rgba_host, frame_dev, bytes, hipMemcpyDeviceToHost));
  else HIP_TRY(hipMemcpy(frame_dev, rgba_host, bytes, hipMemcpyHostToDevice));
  return 0;
}
int ezrt_frame_read(const float* frame_dev, int width, int height, float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  return frame_copy(const_cast<float*>(frame_dev), rgba_host, (size_t)width * height * sizeof(float4), true);
}
int ezrt_frame_write(float* frame_dev, int width, int height, const float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  return frame_copy(frame_dev, const_cast<float*>(rgba_host), (size_t)width * height * sizeof(float4), false);
}

int ezrt_frame_nonfinite(const float* frame_dev, int width, int height, void* stream, int64_t* n_pixels) {
  if (!frame_dev || !n_pixels || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* cnt = nullptr;
  HIP_TRY(hipMalloc((void**)&cnt, sizeof *cnt));
  hipError_t e = hipMemsetAsync(cnt, 0, sizeof *cnt, st);
  const size_t n = (size_t)width * (size_t)height;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(frame_dev), n, cnt);
    e = hipGetLastError();
  }
  unsigned long long host = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&host, cnt, sizeof host, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(cnt);
  if (e != hipSuccess) return fail(EZRT_ERR_DEVICE, "ezrt_frame_nonfinite: %s", hipGetErrorString(e));
  *n_pixels = (int64_t)host;
  return 0;
}

int ezrt_render_paths(EzrtScene* s, const EzrtRenderParams* p, int32_t* tri_id, float* t_hit, float* colour) {
  if (s) (void)hipDeviceSynchronize(); // (a synchronous audit call: pipelined chunks of earlier render calls may still own the scratch it reuses)
  if (!s || !tri_id || !t_hit) return fail(EZRT_ERR_INVALID, "NULL argument");
  int rc = validate_params(s, p);
  if (rc) return rc;
  rc = build_blocks(s, *p, nullptr);
  if (rc) return rc;
  const int nb = (int)s->blocks_host.size();
  const size_t npix = (size_t)p->width * p->height;
  const int slots = 1 + 2 * p->max_bounce;
  DevBuf<int32_t> dtri;
  DevBuf<float> dt, dcol;
  HIP_TRY(dtri.ensure(npix * slots));
  HIP_TRY(dt.ensure(npix * slots));
  HIP_TRY(dcol.ensure(npix * 3));
  HIP_TRY(hipMemcpy(dtri.p, tri_id, npix * slots * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dt.p, t_hit, npix * slots * sizeof(float), hipMemcpyHostToDevice));
  if (colour) HIP_TRY(hipMemcpy(dcol.p, colour, npix * 3 * sizeof(float), hipMemcpyHostToDevice));
  if (nb > 0 && s->tune.audit_via_queue) {
    // the timed pipeline (raygen -> traceq_kernel + redo -> shading stages), one frame, hit records logged per stage
    rc = ensure_events(s);
    if (rc) return rc;
    s->timed = false;
    s->n_trace_events = 0;
    s->n_trace_launches = 0;
    Pipe& q = s->pipe[0];
    HIP_TRY(q.samples.ensure((size_t)nb * BLOCK));
    PathLogTarget tgt;
    tgt.tri = dtri.p;
    tgt.t = dt.p;
    tgt.colour = colour ? dcol.p : nullptr;
    rc = wavefront_chunk(s, q, p, nb, p->frame0, 1u, nullptr, &tgt);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
  } else if (nb > 0) {
    TraceArgs a;
    a.sc = s->dev();
    a.p = *p;
    a.blocks = s->blocks.p;
    a.n_blocks = nb;
    a.frame_first = p->frame0;
    a.samples = nullptr;
    a.counters = s->counters.p;
    a.log_tri = dtri.p;
    a.log_t = dt.p;
    a.log_colour = dcol.p;
    a.stack_entries = s->depth;
    launch_trace(a, 2, dim3((unsigned)nb), stack_lds_bytes(s), nullptr);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(tri_id, dtri.p, npix * slots * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t_hit, dt.p, npix * slots * sizeof(float), hipMemcpyDeviceToHost));
  if (colour) HIP_TRY(hipMemcpy(colour, dcol.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ezrt_query_hits(EzrtScene* s, const float* rays, int n_rays, int32_t* tri_id, float* t_hit) {
  if (!s || !rays || !tri_id || !t_hit || n_rays < 0) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (n_rays == 0) return 0;
  DevBuf<float> dr, dt;
  DevBuf<int32_t> dtri;
  HIP_TRY(dr.ensure((size_t)n_rays * 6));
  HIP_TRY(dt.ensure((size_t)n_rays));
  HIP_TRY(dtri.ensure((size_t)n_rays));
  HIP_TRY(hipMemcpy(dr.p, rays, (size_t)n_rays * 6 * sizeof(float), hipMemcpyHostToDevice));
  if (s->tune.audit_via_queue) {
    // the rays as ONE stage of a render call: same kernel template, LDS layout, pools, stealing, redo launch
    HIP_TRY(hipDeviceSynchronize()); // (pipelined chunks of earlier render calls may still own the scratch this reuses)
    int rc = ensure_num_cus(s);
    if (rc) return rc;
    Pipe& pp = s->pipe[0];
    const size_t n = (size_t)n_rays;
    constexpr size_t HEAD_SLOT = (size_t)TRACE_HEADS * TRACE_HEAD_STRIDE;
    HIP_TRY(pp.rq_o[0].ensure(n));
    HIP_TRY(pp.rq_d[0].ensure(n));
    HIP_TRY(pp.hits2[0].ensure(n));
    HIP_TRY(pp.redo_slots.ensure(n));
    if (pp.redo_flag.n < n) {
      HIP_TRY(pp.redo_flag.ensure(n));
      HIP_TRY(hipMemset(pp.redo_flag.p, 0, pp.redo_flag.n * sizeof(uint32_t)));
    }
    HIP_TRY(pp.qheads.ensure(QHEADS_WORDS));
    HIP_TRY(hipMemset(pp.qheads.p, 0, QHEADS_WORDS * sizeof(uint32_t)));
    HIP_TRY(pp.qcounts.ensure(320));
    HIP_TRY(hipMemset(pp.qcounts.p, 0, 320 * sizeof(uint32_t)));
    const unsigned g1 = (unsigned)((n + 255) / 256);
    // timing events as for a render call: ezrt_last_render_ms then reports this query (total = pack .. unpack, trace = the
    // stage's trace + redo launches) -- how ray-order experiments time the TIMED kernel on caller-chosen rays
    rc = ensure_events(s);
    if (rc) return rc;
    s->timed = false;
    s->n_trace_events = 0;
    s->n_trace_launches = 0;
    HIP_TRY(hipEventRecord(s->ev_begin, nullptr));
    hipLaunchKernelGGL(query_pack_kernel, dim3(g1), dim3(256), 0, nullptr, dr.p, (uint32_t)n, pp.rq_o[0].p, pp.rq_d[0].p, pp.qcounts.p);
    const TraceCfg cfg = trace_cfg(s);
    const bool shared_origin = s->tune.audit_via_queue >= 2;
    TraceQArgs t;
    t.sc = trace_scene(s->dev());
    t.rq.o = pp.rq_o[0].p;
    t.rq.d = pp.rq_d[0].p;
    t.hits = pp.hits2[0].p;
    t.n_paths = pp.qcounts.p;
    t.rays_per_path = 1u;
    t.const_origin = shared_origin ? 1u : 0u;
    t.inner_rel = nullptr;
    t.origin[0] = rays[0];
    t.origin[1] = rays[1];
    t.origin[2] = rays[2];
    const bool wide = use_wide4(s);
    const float4* rel4 = nullptr;
    if (shared_origin && s->tune.rel_boxes && wide) {
      HIP_TRY(pp.inner4_rel.ensure((size_t)s->n_inner4 * N4_FLOAT4));
      hipLaunchKernelGGL(inner4_rel_kernel, dim3((unsigned)((s->n_inner4 + 255) / 256)), dim3(256), 0, nullptr, s->inner4.p,
                         s->n_inner4, rays[0], rays[1], rays[2], pp.inner4_rel.p);
      rel4 = pp.inner4_rel.p;
    } else if (shared_origin && s->tune.rel_boxes && s->n_inner > 0) {
      HIP_TRY(pp.inner_rel.ensure((size_t)s->n_inner * 4));
      hipLaunchKernelGGL(inner_rel_kernel, dim3((unsigned)((s->n_inner + 255) / 256)), dim3(256), 0, nullptr, s->inner.p, s->n_inner,
                         rays[0], rays[1], rays[2], pp.inner_rel.p);
      t.inner_rel = pp.inner_rel.p;
    }
    t.head = pp.qheads.p;
    t.counters = s->counters.p;
    fill_trace_knobs(s, cfg, t);
    t.dbg = nullptr;
    t.slot_map = nullptr;
    t.steal = s->tune.steal ? 1u : 0u;
    t.count_rays = 1u;
    t.redo_count = pp.qcounts.p + 128;
    t.redo_slots = pp.redo_slots.p;
    t.redo_flag = pp.redo_flag.p;
    t.force_pending = 0u;
    t.wave_log = nullptr;
    HIP_TRY(hipEventRecord(s->ev_trace[0][0], nullptr));
    if (wide) launch_traceq4_cfg(s, trace_cfg4(s, rel4 != nullptr), t, rel4, nullptr, nullptr);
    else launch_traceq_cfg(s, cfg, t, false, nullptr);
    if (t.steal || wide) {
      TraceQArgs r = t;
      r.steal = 0u;
      r.count_rays = 0u;
      r.slot_map = pp.redo_slots.p;
      r.n_paths = pp.qcounts.p + 128;
      r.head = pp.qheads.p + (size_t)40 * HEAD_SLOT;
      launch_traceq_cfg(s, cfg, r, true, nullptr);
    }
    HIP_TRY(hipEventRecord(s->ev_trace[0][1], nullptr));
    s->n_trace_events = 1;
    hipLaunchKernelGGL(query_unpack_kernel, dim3(g1), dim3(256), 0, nullptr, pp.hits2[0].p, (uint32_t)n, dtri.p, dt.p);
    HIP_TRY(hipEventRecord(s->ev_end, nullptr));
    s->timed = true;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(tri_id, dtri.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t_hit, dt.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  QueryArgs a;
  a.sc = s->dev();
  a.rays = dr.p;
  a.n = n_rays;
  a.tri = dtri.p;
  a.t = dt.p;
  a.counters = s->counters.p;
  dim3 grid((unsigned)((n_rays + BLOCK - 1) / BLOCK));
  if (s->instr > 0) hipLaunchKernelGGL(query_kernel<true>, grid, dim3(BLOCK), stack_lds_bytes(s), nullptr, a);
  else hipLaunchKernelGGL(query_kernel<false>, grid, dim3(BLOCK), stack_lds_bytes(s), nullptr, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(tri_id, dtri.p, (size_t)n_rays * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t_hit, dt.p, (size_t)n_rays * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ezrt_tonemap(const float* rgba, int n_pixels, uint8_t* rgb8) {
  if (!rgba || !rgb8 || n_pixels < 0) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (n_pixels == 0) return 0;
  DevBuf<float4> din;
  DevBuf<uint8_t> dout;
  HIP_TRY(din.ensure((size_t)n_pixels));
  HIP_TRY(dout.ensure((size_t)n_pixels * 3));
  HIP_TRY(hipMemcpy(din.p, rgba, (size_t)n_pixels * sizeof(float4), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, nullptr, din.p, n_pixels,
                     dout.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(rgb8, dout.p, (size_t)n_pixels * 3, hipMemcpyDeviceToHost));
  return 0;
}

int ezrt_sobol(uint32_t index0, int n, int n_dims, float* out) {
  if (!out || n < 0 || n_dims < 1 || n_dims > 16) return fail(EZRT_ERR_INVALID, "bad sobol arguments");
  if (n == 0) return 0;
  DevBuf<float> d;
  size_t cnt = (size_t)n * n_dims;
  HIP_TRY(d.ensure(cnt));
  hipLaunchKernelGGL(sobol_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, nullptr, index0, n, n_dims, d.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, d.p, cnt * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ezrt_scene_set_sampler(EzrtScene* s, int sobol_dims) {
  if (!s || (sobol_dims != 8 && sobol_dims != 16)) return fail(EZRT_ERR_INVALID, "sobol_dims must be 8 or 16");
  s->sobol_mask = (uint32_t)sobol_dims - 1u;
  return 0;
}

int ezrt_set_option(EzrtScene* s, const char* name, int value) {
  if (!s || !name) return fail(EZRT_ERR_INVALID, "NULL argument");
  for (const TuningName& k : kTuning)
    if (strcmp(k.name, name) == 0) {
      if (value < k.lo || value > k.hi)
        return fail(EZRT_ERR_INVALID, "option '%s' = %d outside [%d, %d]", name, value, k.lo, k.hi);
      s->tune.*(k.field) = value;
      return 0;
    }
  return fail(EZRT_ERR_INVALID, "unknown option '%s'", name);
}

int ezrt_set_instrumentation(EzrtScene* s, int level) {
  if (!s || level < 0 || level > 1) return fail(EZRT_ERR_INVALID, "bad instrumentation level");
  s->instr = level;
  return 0;
}
int ezrt_counters(EzrtScene* s, uint64_t out[EZRT_CTR_COUNT]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long all[CTR_SLOTS * EZRT_CTR_COUNT];
  HIP_TRY(hipMemcpy(all, s->counters.p, sizeof all, hipMemcpyDeviceToHost));
  for (int k = 0; k < EZRT_CTR_COUNT; k++) {
    out[k] = 0;
    for (int j = 0; j < CTR_SLOTS; j++) out[k] += all[j * EZRT_CTR_COUNT + k];
  }
  return 0;
}
int ezrt_counters_reset(EzrtScene* s) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(s->counters.p, 0, (size_t)CTR_SLOTS * EZRT_CTR_COUNT * sizeof(unsigned long long)));
  return 0;
}
int ezrt_last_render_ms(EzrtScene* s, float* total_ms, float* trace_kernel_ms, int* n_trace_launches) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (!s->timed) return fail(EZRT_ERR_INVALID, "no render call to time yet");
  HIP_TRY(hipEventSynchronize(s->ev_end));
  float tot = 0.0f, tr = 0.0f;
  HIP_TRY(hipEventElapsedTime(&tot, s->ev_begin, s->ev_end));
  for (int i = 0; i < s->n_trace_events; i++) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_trace[i][0], s->ev_trace[i][1]));
    tr += ms;
  }
  if (total_ms) *total_ms = tot;
  if (trace_kernel_ms) *trace_kernel_ms = tr;
  if (n_trace_launches) *n_trace_launches = s->n_trace_launches;
  return 0;
}
int ezrt_scene_prune_info(EzrtScene* s, double out[8]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  out[0] = s->prunable ? (double)prune_mode(s) : -1.0;
  out[1] = s->prune_G;
  out[2] = s->prune_Z;
  out[3] = s->prune_M;
  out[4] = (double)s->prune_bad;
  out[5] = (double)s->prune_a;
  out[6] = s->retreed ? 1.0 : 0.0;
  out[7] = (double)s->n_inner4;
  return 0;
}
int ezrt_scene_stats(EzrtScene* s, int64_t out[6]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  memcpy(out, s->stats, sizeof s->stats);
  return 0;
}

int ezrt_debug_math(int op, const float* a, const float* b, int n, float* out) {
  if (!a || !out || n < 0 || op < 0 || op > 18) return fail(EZRT_ERR_INVALID, "bad argument");
  if (n == 0) return 0;
  if (op == 18) { // exhaustive audit of the device's correctly rounded reciprocal (ez_rcp): out[0] = mismatches over all 2^32 inputs, out[1] = bits of the first
    if (n < 2) return fail(EZRT_ERR_INVALID, "op 18 writes two values");
    DevBuf<unsigned long long> res;
    HIP_TRY(res.ensure(2));
    const unsigned long long init[2] = {0ull, ~0ull};
    HIP_TRY(hipMemcpy(res.p, init, sizeof init, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rcp_audit_kernel, dim3(4096), dim3(256), 0, nullptr, res.p);
    HIP_TRY(hipGetLastError());
    unsigned long long got[2];
    HIP_TRY(hipMemcpy(got, res.p, sizeof got, hipMemcpyDeviceToHost));
    out[0] = (float)(got[0] > 16777216ull ? 16777216ull : got[0]);
    const uint32_t fb = got[0] ? (uint32_t)got[1] : 0u;
    memcpy(&out[1], &fb, 4);
    for (int i = 2; i < n; i++) out[i] = 0.0f;
    return 0;
  }
  if (op == 17) { // floor(bits(a[i]) / bits(b[0])) through the kernels' FastDiv
    uint32_t d = 0;
    if (!b) return fail(EZRT_ERR_INVALID, "bad argument");
    memcpy(&d, b, 4);
    if (d == 0) return fail(EZRT_ERR_INVALID, "division by zero");
    DevBuf<float> da, dout;
    HIP_TRY(da.ensure((size_t)n));
    HIP_TRY(dout.ensure((size_t)n));
    HIP_TRY(hipMemcpy(da.p, a, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fastdiv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, da.p, make_fastdiv(d), n, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  // ops 10-12 (intersector audit): a = n rays of 6 floats, b = n boxes of 6 / triangles of 9 floats;
  // ops 13-16 (integrator 52's sampler): a = n x 6, b = n x 6 material parameters
  const size_t wa = op >= 10 ? 6 : 1, wb = op == 11 ? 9 : (op >= 10 ? 6 : 1);
  if (op >= 10 && !b) return fail(EZRT_ERR_INVALID, "bad argument");
  DevBuf<float> da, db, dout;
  HIP_TRY(da.ensure((size_t)n * wa));
  HIP_TRY(db.ensure((size_t)n * wb));
  HIP_TRY(dout.ensure((size_t)n));
  HIP_TRY(hipMemcpy(da.p, a, (size_t)n * wa * sizeof(float), hipMemcpyHostToDevice));
  if (b) HIP_TRY(hipMemcpy(db.p, b, (size_t)n * wb * sizeof(float), hipMemcpyHostToDevice));
  else HIP_TRY(hipMemset(db.p, 0, (size_t)n * sizeof(float)));
  if (op >= 10)
    hipLaunchKernelGGL(isect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, da.p, db.p, n, dout.p);
  else
    hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, da.p, db.p, n, dout.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

} // extern "C"
