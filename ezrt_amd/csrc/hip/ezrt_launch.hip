// ezrt_launch.hip -- every kernel launch of libezrt_hip.so: the launch policy of a render call (which instance of which kernel, LDS
// budgets and workgroups per CU, chunks, the two scratch sets and their streams), the audit entry points (ezrt_render_paths,
// ezrt_query_hits, ezrt_debug_math) and the small utility kernels (tone map, Sobol, non-finite count).  DESIGN.md 5.
#include "ezrt_internal.h"
#include "ezrt_kernels.h"
#include "ezrt_wavefront.h"
#include "ezrt_traceq4.h"

namespace {

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
// stack rows of a traceq4 launch: the exact worst case of the slot-order traversal (prune 0 / 1); the nearest-first order (prune 2)
// has no small bound and runs on a ring with a spill area (below)
// (round 6) prune 2: the rows are a RING of stack_cap4 rows -- a power of two, 16 unless the knob stack_cap says less -- + one row of
// spill counters; entries beyond the ring go to the lane's spill area in global memory (TraceQ4Args::stack_cap), so the launch no
// longer allocates the exact worst case (C3 23, C5 24 rows) for stacks that use 13-17 rows at most
constexpr int OVF_CAP = 16; // spilled entries per lane (ring + spill = 32 deep before a ray is handed to the redo list)
int stack_cap4(const EzrtScene* s) { // (prune 2 only: the other modes run with their exact bound)
  int want = s->tune.stack_cap > 0 ? s->tune.stack_cap : 16;
  if (s->tune.debug_stack_cap > 0) want = 4; // (test hook: a tiny ring, and a spill area of 4 (debug_stack_cap - 1) entries)
  int r = 4;
  while (r * 2 <= want && r < 32) r *= 2;
  return r;
}
int ovf_cap4(const EzrtScene* s) { return s->tune.debug_stack_cap > 0 ? 4 * (s->tune.debug_stack_cap - 1) : OVF_CAP; }
int stack_rows4(const EzrtScene* s) {
  if (prune_mode(s) != 2) return s->stack_need4;
  // (knob prune_mis != 2: the two-ray launches then run a slot-order instance on ABSOLUTE rows -- they need the exact bound too)
  return s->tune.prune_mis != 2 ? std::max(stack_cap4(s) + 1, s->stack_need4) : stack_cap4(s) + 1;
}
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
    // (the bounce stages at seven waves per SIMD -- 72 VGPRs, 24 spills -- were measured once the ring stack had freed the LDS for a
    // seventh workgroup: C2 -7 %, C3 -6 %, C4 -4 %, C5 -7 %; profiles/r6/ring_stack_ab.txt)
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
// ovf: the scratch set's spill area of the traversal stacks (Pipe::ovf: [grid lanes][OVF_CAP])
void fill_traceq4_args(const EzrtScene* s, const TraceCfg& c4, const TraceQArgs& t, const float4* rel, const WfArgs* gen, uint32_t* ovf, TraceQ4Args& A) {
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
    A.stack_cap = stack_cap4(s);
    A.ovf = ovf;
    A.ovf_cap = ovf ? ovf_cap4(s) : 0;
  }
  // (the even / odd slots of a two-ray path must stay in one granule: any granule >= 2 slots does)
  // knob bounce_scatter: 1 (default) = queues with one ray per path only.  Measured in the pipeline (profiles/r4/bounce_scatter_ab.txt):
  // C2 +2.7 % (trace launches 1.39 -> 1.345 ms), C3 -0.5 % (noise); the MIS integrators' queues (two rays per path sharing an
  // origin, env shadow rays that are coherent by construction) LOST 1.3 % (C4) and 2.8 % (C5) with it: never scattered
  A.gscat_shift = (!rel && !gen && !t.slot_map && s->tune.bounce_scatter != 0 && t.rays_per_path == 1u) ? 3u : 0u;
  A.handover = (s->tune.handover && t.steal) ? 1u : 0u;
  A.steal_bound = s->tune.steal_bound ? 1u : 0u;
}
void launch_traceq4_cfg(EzrtScene* s, const TraceCfg& c4, const TraceQArgs& t, const float4* rel, hipStream_t st, const WfArgs* gen, uint32_t* ovf) {
  TraceQ4Args A;
  fill_traceq4_args(s, c4, t, rel, gen, ovf, A);
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
template <int INTEG, int STAGE>
void launch_shade_split_ib(const WfArgs& a, dim3 grid, dim3 grid_hit, hipStream_t st) {
  hipLaunchKernelGGL((shade_miss_kernel<INTEG, false, STAGE>), grid, dim3(SHADE_BLOCK), 0, st, a);
  hipLaunchKernelGGL((shade_hit_kernel<INTEG, false, STAGE>), grid_hit, dim3(SHADE_BLOCK), 0, st, a);
}
template <int INTEG>
void launch_shade_i(const WfArgs& a, bool full, dim3 grid, hipStream_t st) {
  if (!full && a.bounce == 0) launch_shade_split_ib<INTEG, 0>(a, grid, grid, st);
  else if (!full && a.bounce == 1) launch_shade_split_ib<INTEG, 1>(a, grid, grid, st);
  else launch_shade_fused_i<INTEG>(a, full, grid, st);
}
// (The redo launch of a stage -- exact ties beyond two candidates, rays that are not tame, spill areas that ran full: normally EMPTY -- runs in
// line before any shading.  Until round 6 a knob could put it on a side stream under the first shading pass: the two measured the same, and the
// cross-stream waits it needed can enter a slow state on this runtime (ezrt_streams.h); removed with its stream and events.)
void launch_shade(const WfArgs& a, bool full, dim3 grid, hipStream_t st) {
  switch (a.p.integrator) {
    case EZRT_INTEGRATOR_P3_DIFFUSE: launch_shade_i<EZRT_INTEGRATOR_P3_DIFFUSE>(a, full, grid, st); break;
    case EZRT_INTEGRATOR_P4_DISNEY: launch_shade_i<EZRT_INTEGRATOR_P4_DISNEY>(a, full, grid, st); break;
    case EZRT_INTEGRATOR_P5_SOBOL: launch_shade_i<EZRT_INTEGRATOR_P5_SOBOL>(a, full, grid, st); break;
    case EZRT_INTEGRATOR_P5_MIS_ANISO: launch_shade_i<EZRT_INTEGRATOR_P5_MIS_ANISO>(a, full, grid, st); break;
    default: launch_shade_i<EZRT_INTEGRATOR_P5_MIS>(a, full, grid, st); break;
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

void release_chunk_scratch_impl(Pipe& pp) {
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
  HIP_TRY(pp.ovf.ensure((size_t)s->num_cus * 8 * BLOCK * OVF_CAP)); // (spill area of the traversal stacks: at most 8 workgroups per CU)
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
        launch_traceq4_cfg(s, rel ? cfg4_rel : cfg4_abs, t, rel ? pp.inner4_rel.p : nullptr, st, (rel && gen_primary) ? &a : nullptr, pp.ovf.p);
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
        launch_traceq(r, true);
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
    launch_shade(a, full, dim3(shade_grid), st);
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
        std::vector<double> endt, life, its, rays, startt, exht, after, maxsp;
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
            maxsp.push_back((double)(w[i * 8 + 5] >> 48));
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
        fprintf(stderr, "[ezrt]   refill block in %.0f %% of the iterations, steal block in %.0f %% | highest stack row used by a wave: p50 %.0f p99 %.0f max %.0f\n",
                100.0 * (double)s_rf / (double)(s_it ? s_it : 1), 100.0 * (double)s_st / (double)(s_it ? s_it : 1), pct(maxsp, 0.5), pct(maxsp, 0.99), pct(maxsp, 1.0));
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

namespace ezi {
void release_chunk_scratch(Pipe& pp) { ::release_chunk_scratch_impl(pp); }
} // namespace ezi

extern "C" {

static int ezrt_render_device_body(EzrtScene* s, const EzrtRenderParams* p, float* accum_dev, void* stream) {
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
        for (Pipe& q : s->pipe) release_chunk_scratch_impl(q);
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
int ezrt_render_device(EzrtScene* s, const EzrtRenderParams* p, float* accum_dev, void* stream) {
  return ezi::guarded("ezrt_render_device", [&]() -> int { return ezrt_render_device_body(s, p, accum_dev, stream); });
}

static int ezrt_render_body(EzrtScene* s, const EzrtRenderParams* p, float* accum) {
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
int ezrt_render(EzrtScene* s, const EzrtRenderParams* p, float* accum) {
  return ezi::guarded("ezrt_render", [&]() -> int { return ezrt_render_body(s, p, accum); });
}

static int ezrt_frame_nonfinite_body(const float* frame_dev, int width, int height, void* stream, int64_t* n_pixels) {
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
int ezrt_frame_nonfinite(const float* frame_dev, int width, int height, void* stream, int64_t* n_pixels) {
  return ezi::guarded("ezrt_frame_nonfinite", [&]() -> int { return ezrt_frame_nonfinite_body(frame_dev, width, height, stream, n_pixels); });
}

static int ezrt_render_paths_body(EzrtScene* s, const EzrtRenderParams* p, int32_t* tri_id, float* t_hit, float* colour) {
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
int ezrt_render_paths(EzrtScene* s, const EzrtRenderParams* p, int32_t* tri_id, float* t_hit, float* colour) {
  return ezi::guarded("ezrt_render_paths", [&]() -> int { return ezrt_render_paths_body(s, p, tri_id, t_hit, colour); });
}

static int ezrt_query_hits_body(EzrtScene* s, const float* rays, int n_rays, int32_t* tri_id, float* t_hit) {
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
    if (wide) {
      HIP_TRY(pp.ovf.ensure((size_t)s->num_cus * 8 * BLOCK * OVF_CAP));
      launch_traceq4_cfg(s, trace_cfg4(s, rel4 != nullptr), t, rel4, nullptr, nullptr, pp.ovf.p);
    }
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
int ezrt_query_hits(EzrtScene* s, const float* rays, int n_rays, int32_t* tri_id, float* t_hit) {
  return ezi::guarded("ezrt_query_hits", [&]() -> int { return ezrt_query_hits_body(s, rays, n_rays, tri_id, t_hit); });
}

static int ezrt_tonemap_body(const float* rgba, int n_pixels, uint8_t* rgb8) {
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
int ezrt_tonemap(const float* rgba, int n_pixels, uint8_t* rgb8) {
  return ezi::guarded("ezrt_tonemap", [&]() -> int { return ezrt_tonemap_body(rgba, n_pixels, rgb8); });
}

static int ezrt_sobol_body(uint32_t index0, int n, int n_dims, float* out) {
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
int ezrt_sobol(uint32_t index0, int n, int n_dims, float* out) {
  return ezi::guarded("ezrt_sobol", [&]() -> int { return ezrt_sobol_body(index0, n, n_dims, out); });
}

static int ezrt_debug_math_body(int op, const float* a, const float* b, int n, float* out) {
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
int ezrt_debug_math(int op, const float* a, const float* b, int n, float* out) {
  return ezi::guarded("ezrt_debug_math", [&]() -> int { return ezrt_debug_math_body(op, a, b, n, out); });
}

} // extern "C"
