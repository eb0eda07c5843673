// ezrt_internal.h -- what the three translation units of libezrt_hip.so share (round 6: ezrt_hip.hip was one 2 800-line file):
//   ezrt_hip.hip          the C ABI's scene lifetime, options, counters, frames and error plumbing
//   ezrt_scene_build.hip  ezrt_scene_create / ezrt_scene_set_env: the device layout of a scene -- records, the re-tree over the
//                         reference's leaves, the 4-wide collapse, the per-triangle pruning bounds
//   ezrt_launch.hip       every kernel launch: the launch policy of a render call (variants, LDS budgets, chunks, streams), the audit
//                         entry points and the small utility kernels
// Types only -- no kernel is defined here (a __global__ function may live in one translation unit only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "ezrt.h"
#include "ezrt_device.h"
#include "ezrt_records.h"
#include "ezrt_streams.h"

using namespace ezd;

namespace ezi {
// error plumbing (defined in ezrt_hip.hip): the message of the calling thread's last failed entry point
__attribute__((visibility("hidden"))) int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
__attribute__((visibility("hidden"))) const char* last_error();
} // namespace ezi
using ezi::fail;

namespace ezi {
// No exception crosses the C ABI: every entry point that can allocate host memory (std::vector, std::string, worker threads) runs
// its body through this (round 6: four hand-written catch sites guarded ~40 entry points).
template <class F>
inline int guarded(const char* what, F body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(EZRT_ERR_NOMEM, "%s: out of host memory", what);
  } catch (const std::exception& e) {
    return fail(EZRT_ERR_DEVICE, "%s failed: %s", what, e.what());
  } catch (...) {
    return fail(EZRT_ERR_DEVICE, "%s failed: unknown exception", what);
  }
}
} // namespace ezi

namespace {


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
  DevBuf<uint32_t> ovf;         // traceq4_kernel's spill area of the LDS traversal stacks: [launch lanes][OVF_CAP] (TraceQ4Args::stack_cap)
  DevBuf<uint32_t> qheads;      // traceq reservation counters: [launch slot][TRACE_HEADS][TRACE_HEAD_STRIDE] (QHEADS_WORDS in all; zeroed per chunk)
  DevBuf<float4> inner_rel;     // inner records translated by -eye (primary rays)
  DevBuf<float4> inner4_rel;    // 4-wide records translated by -eye
  DevBuf<uint4> defer_list;     // split shading: paths with a surface interaction, per workgroup
  DevBuf<uint32_t> defer_count;
  int stream_device = 0;         // device `stream` belongs to
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

// distance pruning of the timed stages: knob and scene property (ezrt_traceq4.h "Distance pruning")
inline int prune_mode(const EzrtScene* s) {
  if (!s->prunable || s->n_inner4 < s->tune.prune_min_records) return 0;
  return s->tune.prune;
}

namespace ezi {
// ezrt_launch.hip
__attribute__((visibility("hidden"))) void release_chunk_scratch(Pipe& pp);
} // namespace ezi
