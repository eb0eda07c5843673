// ezrt_traceq4.h -- traceq4_kernel: the persistent hitBVH of ezrt_traceq.h over a 4-WIDE collapse of the
// reference's binary tree (the dominant kernel of a render call since round 2).
//
// Why this is still the reference's hitBVH (P5/fsh:254-306), bit for bit:
//
//  * hitBVH does not prune: a node is visited iff the slab test of its own box and of every ancestor's box
//    (below the root) says "hit" (d > 0).  For a ray whose origin and 1/direction are finite ("tame") the slab
//    test is MONOTONE in the box: fl(x - S) and fl(. * inv) are monotone, so for a box G nested in a box C
//    (AA_C <= AA_G, BB_G <= BB_C componentwise) every per-axis entry distance of G is >= C's and every exit
//    distance <= C's, hence t0_G >= t0_C, t1_G <= t1_C and  hit(G) = (t1 >= t0 && t1 > 0)  implies hit(C).
//    (hit(.) is exactly the reference's `hitAABB(..) > 0`: with t1 >= t0 it returns t0 if t0 > 0, else t1.)
//    The builders make every child box from a subset of its parent's triangles, so boxes ARE nested;
//    ezrt_scene_create verifies that for caller-supplied arrays and otherwise keeps the binary kernel.
//    So a node deep in the tree can be tested DIRECTLY, without its ancestors: a 4-wide record holds the
//    boxes of a "cut" of up to four descendants of one binary node (children, or grandchildren, ...), and a
//    ray visits exactly the same leaves -- it tests fewer boxes and needs less than half the dependent
//    memory round trips (C2: 0.43 record visits per binary inner visit).
//  * The set of triangles tested is unchanged -- or, with distance pruning (below), reduced by triangles that provably
//    cannot win or tie -- and the result is the minimum of t over that set; the reference's visit order only decides
//    exact ties in t.  This kernel visits slots nearest-first (PRUNE == 2) or in slot order, never in the reference's
//    order, so an exact tie at the running minimum is ordered explicitly: two candidates by two hitAABB calls at the lowest
//    common ancestor of their leaves in the REFERENCE's tree (tie_precedes); a third candidate, a tie between the
//    contributors of a split ray, or missing tables send the ray to the redo list, which the in-order binary kernel re-traces.
//  * Since round 3 the records are usually NOT a collapse of the reference's inner nodes but of the library's own binned-SAH
//    tree over the reference's LEAVES (retree_leaves, ezrt_hip.hip): by the first bullet a leaf is reached iff its own box
//    is hit, whatever the inner boxes above it are, as long as they are unions of leaf boxes.
//  * Rays with a direction component that is exactly +-0 ("semi": 1/d = +-inf there, everything else finite) are common in
//    chapter 5 -- SampleHdr's phi is exactly 0 for every cache cell whose x is 0.5, so whole families of env shadow rays have
//    L.z = 0 -- and each used to walk the tree in ONE lane of the in-order kernel (C5: 10 % of a frame in redo launches).
//    Their slab products are finite or +-inf and ordered exactly like a tame ray's (the select form picks the near / far
//    row by the sign of +-inf; an interval nested in another keeps its products inside the other's in the extended reals),
//    EXCEPT 0 x inf = NaN when a box plane coincides with the origin on that axis, which hardware min3 / max3 would drop
//    silently: waves holding such a ray test their products for NaN (used slots only: unused slots are NaN on purpose) and
//    hand the ray to the redo list at the first one.  Their pruning margin is infinite (max |1/d| = inf): never pruned.
//  * Rays that are not tame in any other way (non-finite origin or direction) go to the redo list unseen.
//
// Record (128 B in HBM = one L2 line, 112 B in LDS): AAx[4] AAy[4] AAz[4] BBx[4] BBy[4] BBz[4] ref[4] (pad);
// an unused slot has an all-NaN box (v_min3/v_max3 of three NaNs is NaN and every compare with it is false:
// never hit, no extra instruction) and ref = REF_EMPTY.  Slots are ordered by ascending stack need of their
// subtrees; without the nearest-first order (PRUNE < 2) they are visited lowest-first with the others pushed in descending
// order, which bounds the LDS stack by max_j (pending_j + need_j) -- 16 rows on C2 where the binary traversal needs 18.  The
// nearest-first order (PRUNE == 2, the default) has no such bound: its rows are a ring of 16 with a spill area in global memory
// (TraceQ4Args::stack_cap; round 6), and only a lane whose spill area is full hands its ray to the redo list.
//
// Everything else (persistent waves, prefetched next ray, batched refill, static + dynamic pools, postponed
// cooperative leaves, intra-wave stealing with the 64-bit atomicMin merge) is ezrt_traceq.h's schedule.
#pragma once
#include "ezrt_records.h"
#include "ezrt_traceq.h"

namespace ezd {

// The near / far plane of each axis is SELECTED by the sign of 1/direction -- through the address of the 16-byte row that is
// loaded -- instead of computed with v_min / v_max of both products.  For a tame ray the two are the same numbers: AA <= BB
// componentwise, fl(x - S) and fl(. * inv) are monotone, so with inv >= 0 the AA product is the smaller one and with inv < 0
// the BB product (equal products: either; +-0 only ever feed comparisons; NaN rows of unused slots stay NaN).
// v_min/v_max/v_min3 issue at about half the rate of v_mul/v_sub on gfx950 (profiles/r2/valu_issue_microbench.txt): 24 of them
// per record became 12 integer address operations (round 2: +0.9 % on C2, +2.9 % on C5; the min/max form was removed in round 6).
// (row order and reference bits: ezrt_records.h)
struct TraceQ4Args {
  TraceQArgs q;             // queue, pools, counters, redo list, knobs (q.lds_nodes is unused here)
  const float4* inner4;     // 4-wide records, breadth-first
  const float4* inner4_rel; // q.const_origin only (or NULL): the same with every box translated by -origin
  uint32_t root4;           // reference of the root: record 0, or the root leaf
  int32_t lds_nodes4;       // records [0, lds_nodes4) staged in LDS
  // PRUNE > 0 (see "Distance pruning" below): delta(ray) = (prune_a + prune_cs * |S|_inf) * max_k |1/d_k|
  float prune_a, prune_cs;
  // exact ties resolved in place (see tie_precedes): the reference leaf of every triangle and, per reference node,
  // x = parent | depth << 24, y = the parent's binary record | (1 << 31 if this node is the RIGHT child); NULL: ties go to the redo list
  const int32_t* tri_leaf;
  const int2* ref_up;
  // GEN (primary stage only): the queue holds no directions; the ray of queue position q is primary_dir(gen_*, q)
  EzrtRenderParams gen_p;
  const int2* gen_blocks;
  FastDiv gen_div_blocks, gen_div_sub;
  uint32_t gen_scatter, gen_scatter_shift, gen_frame_first;
  // PRUNE == 2 (nearest-first: no small worst-case bound).  Round 6: the LDS rows of a lane are a RING of stack_cap rows (a power of two:
  // position p lives in row p & (stack_cap - 1)); when fewer than three rows are free after a step -- one step pushes at most three -- the
  // lane's OLDEST (up to four) entries move to its spill area in global memory (ovf: ovf_cap entries per lane) and come back, last in first out, when
  // the ring runs empty; only a lane whose spill area is full too hands its ray to the redo list.  Why: the launch used to allocate the
  // exact worst case of the slot-order traversal + 3 rows (C2 19, C3 23, C5 24 KB of LDS per workgroup) while the per-wave logs show rays
  // using 13-17 rows at most (profiles/r6/wide8_pairs_negative.txt "highest stack row"); 16 rows + a counter row fit a sixth workgroup per CU
  // on the deep scenes.  Row stack_cap of the lane's column holds the number of spilled entries.
  int32_t stack_cap;
  uint32_t* ovf;            // [launch lanes][ovf_cap]
  int32_t ovf_cap;
  // Queue positions are DRAWN in a scattered order (bounce stages; 0 = off, else the template parameter GS): granules of 8 consecutive slots,
  // logical granule g -> physical granule (g mod 256) * R + g / 256, R = ceil(granules / 256) -- a transpose, so that the 16
  // granules of a wave's pool come from 16 places spread over a sixteenth of the queue instead of from one run of 128 slots.
  // A shading workgroup writes its survivors as a run (a few hundred rays from a handful of 8x8 pixel blocks), so
  // consecutive slots cost about the same and a wave that drew an expensive run was the launch's tail: measured on
  // synthetic bounce rays (tools/exp_ray_order.py, profiles/r4/ray_order_experiment_*.txt) granules of 8 shuffled are as
  // fast as a fully random order, C2 -15 % trace time, C3 -3.5 %, C5 +-0 -- and SORTING the rays for coherence (origin
  // cell, direction octant, Morton) is 5-20 % SLOWER than the pipeline's order: mixing cheap and expensive rays in every
  // wave is worth more than coherent memory accesses.  Only the order of processing changes: every ray keeps its slot.
  uint32_t gscat_shift;
  // Ray hand-over (knob handover; round 5): once the queue is exhausted, an idle lane takes the PREFETCHED ray of a lane that is still
  // traversing -- a whole ray that has not started, handed over in registers (shuffles): no split, no atomic, nothing speculative.
  // Without it the last two rays of a lane (the one it traverses, the one it holds prefetched) run one after the other while the
  // lanes next to it have nothing to do: "a wave runs on for" 104 us (median) after stage 1's queue is found empty.
  uint32_t handover;
  uint32_t steal_bound;        // a thief (of its own wave) prunes against the victim's best hit so far
};
// ---- Distance pruning (PRUNE > 0): results-neutral, proven, not merely observed.
//
// The reference's hitBVH never prunes (P5/fsh:254-306), so the result of a ray is the minimum of t over every triangle
// whose leaf the box tests let it reach.  A slot may be SKIPPED without changing that minimum -- or the set of
// triangles that tie with it -- if no triangle below it can be accepted with t <= best_t.  Claim: if hit_triangle_t
// accepts triangle T (vertices inside the slot's box B: ezrt_scene_create verifies every leaf and every nesting) with
// distance t, then for every axis k, with eps = 2^-24, s = |S|_inf, m_T = the largest |coordinate| of T's vertices,
//        t  >=  t0_k(B) - (eta_T + 17 eps s) |inv_k| (1 + eps) - 3.1 eps |t0_k| - eps t,
//        eta_T = a_T + 19.8 eps m_T,      a_T = zeta_T + 15.2 eps (diam_T + zeta_T) / sin(theta'_T / 2),
// where t0_k = fl(fl(near_k - S_k) inv_k) is the slab test's entry product, zeta_T = max_i |N.(p_i - p1)| / |N| the
// distance of T's vertices from the plane through p1 with the STORED normal N, theta'_T the smallest angle and diam_T the
// longest edge of T projected along N.  Proof (long form: DESIGN.md 5).  An accepted hit passed the three edge tests on
// the fp32 point P = fl(S + fl(d t)).  (i) P lies within eps (17.5 m_T + 15.8 s) of that plane however small |N.d| is:
// N.(P* - p1) = e_num + eps2 num - t e_den for P* = S + d t -- the rounding errors of the two dot products, the
// subtraction and the division, NOT divided by N.d: an error of t moves P along the ray, and the plane distance is
// measured across it.  (ii) The edge tests only see projections along N; each computed sign is exact up to
// 15.01 eps |edge| |P - p_i| and |P - p_i| is itself of the size of T (fl(P - p_i) has a RELATIVE error), so the
// projection P' lies within 15.05 eps |P - p_i| of each half-plane of the projected triangle T', hence within
// 15.2 eps (diam_T + zeta_T) / sin(theta'/2) of T' -- the wedge at a corner lets a point that far past the tip; this is
// the term a sliver inflates, and why it is kept PER TRIANGLE.  (iii) T' is within zeta_T of T, and T is inside B.  So
// P_k >= AA_k - eta_T - 15.8 eps s (and <= BB_k + ...); finally P_k = fl(S_k + fl(d_k t)) ties t to (P_k - S_k) / d_k
// up to eps (2.02 m_T + 2 a_T + s) |inv_k|.
// eta_T is evaluated per triangle in double precision at scene creation.  Triangles are split into ORDINARY ones,
// eta_T <= 2^-13 max|coordinate| -- or, where that is not small at the scale of the median triangle, the 1 - 2^-10 quantile
// of the bounds -- (everything but slivers and triangles far from the origin: the thin side faces of a flattened box, a
// ground plane of kilometres), whose maximum gives the launch's prune_a = 2 max eta_T (safety factor 2, rounded up; prune_cs = 2 * 17 eps),
// and the rest -- larger eta_T, or no bound at all (sin(theta'/2) < 1e-4, zeta_T > 1e-3 of the shortest edge, |N| not 1):
// every record above such a triangle carries REF_NOPRUNE in the references to it and none of its slots is ever skipped.
// A leaf box that does not hold its triangles switches pruning off for the whole scene.  A slot is skipped iff
//        t0 = max_k t0_k  >  (best_t + (prune_a + prune_cs s) max_k |inv_k|) (1 + 2^-19)          (strict).
// Then every triangle below it has t > best_t: it can neither win nor tie, so the hit record AND the tie detection (redo
// list) are those of the unpruned traversal.  Counters P/I/T/M stay the unpruned reference's: they come from
// traceq_kernel, which does not prune.
// EZRT_PRUNE_NEGATIVE_CONTROL (never defined in the product build): a library variant whose margin is NEGATIVE -- it skips
// slots up to 1e-3 of the distance IN FRONT of the best hit -- so that tools/gpu_negctl.sh can show that
// tests/test_gpu_prune.py does catch a traversal that prunes too much.
#ifdef EZRT_PRUNE_NEGATIVE_CONTROL
constexpr float PRUNE_REL = 1.0f - 1.0f / 1024.0f;
#else
constexpr float PRUNE_REL = 1.0f + 1.0f / 524288.0f;
#endif // 1 + 2^-19 = 1 + 32 eps: covers 3.1 eps |t0| + eps t and the two roundings here

// 4-wide records with every box translated by -S: (AA - S, BB - S), the subtraction hitAABB does per visit
EZD void inner4_translate(const float4* in, int i, float sx, float sy, float sz, float4* out) {
  const float4* r = in + (size_t)i * N4_FLOAT4;
  float4* o = out + (size_t)i * N4_FLOAT4;
  const float s[3] = {sx, sy, sz};
  for (int k = 0; k < 3; k++) {
    const float4 v = r[N4_ROW_AA + k], w = r[N4_ROW_BB + k];
    const float c = s[k];
    o[N4_ROW_AA + k] = make_float4(v.x - c, v.y - c, v.z - c, v.w - c);
    o[N4_ROW_BB + k] = make_float4(w.x - c, w.y - c, w.z - c, w.w - c);
  }
  o[N4_ROW_REF] = r[N4_ROW_REF];
  o[7] = r[7];
}
__global__ void inner4_rel_kernel(const float4* in, int n, float sx, float sy, float sz, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inner4_translate(in, i, sx, sy, sz, out);
}

// A chunk's housekeeping, done by the threads of raygen_kernel on their way (three launches less per chunk -- two
// memsets and inner4_rel_kernel, ~25 us of launch gaps on a 2.3 ms frame): the trace queue heads and stage counters
// zeroed, the eye-relative copy of the 4-wide records (primary stage).  Everything here is read by later launches only.
struct ChunkPrologue {
  uint32_t* zero_a;   // trace queue heads
  uint32_t n_zero_a;
  uint32_t* zero_b;   // stage counters; word 0 belongs to raygen_kernel (stage 0's path count) and is skipped
  uint32_t n_zero_b;
  const float4* inner4; // NULL: no eye-relative copy wanted
  float4* inner4_rel;
  int32_t n_inner4;
  float sx, sy, sz;
};
EZD void chunk_prologue(const ChunkPrologue& g, uint32_t tid, uint32_t n_threads) {
  for (uint32_t k = tid; k < g.n_zero_a; k += n_threads) g.zero_a[k] = 0u;
  for (uint32_t k = tid + 1u; k < g.n_zero_b; k += n_threads) g.zero_b[k] = 0u;
  if (g.inner4)
    for (uint32_t k = tid; k < (uint32_t)g.n_inner4; k += n_threads) inner4_translate(g.inner4, (int)k, g.sx, g.sy, g.sz, g.inner4_rel);
}

// LOG: the per-wave diagnostics of debug_stages=2 (a.wave_log).  A template parameter, not a run-time test: the
// counters and time stamps are loop-carried values, and even never-executed they cost the production kernel registers
// (one 64-bit time stamp turned 4 spills into 7: -2 %).
// GEN (with REL): primary rays are generated in the refill block -- seed, AA jitter, camera rotation, normalize:
// primary_dir, the code raygen_kernel runs -- and stored to the queue from here (the stage's shading passes and a redo
// launch read them there: recomputing the direction in the shading kernels cost them 30 us each on C2, more than the
// 16-byte read).  What goes away is raygen_kernel's launch and this kernel's read of the queue: the stores ride on a
// memory system the VALU-bound traversal leaves idle.
// Which of two triangles with the SAME hit distance does the reference's hitBVH keep?  The first one it finds (strict <,
// P5/fsh:247, 274), i.e. the one whose leaf comes first in its depth-first order: near child first at every node, ties
// right-first (P5/fsh:291-298).  Two leaves are ordered at their lowest common ancestor alone -- the whole subtree of the
// child visited first precedes the other's -- so the answer is two hitAABB calls on that node's children, found by climbing
// from the two leaves (parent + depth per reference node), instead of re-tracing the ray in reference order: the redo
// launches of C5 spent 0.2-1.5 ms per stage walking a few tied rays through the tree, one lane each (10 % of its frame).
// Same leaf: the lower index (hitArray scans upwards).  Only called for rays that are tame.
EZD bool tie_precedes(const TraceQ4Args& A, int32_t tri_a, int32_t tri_b, f3 S, f3 inv) {
  int32_t a = A.tri_leaf[tri_a], b = A.tri_leaf[tri_b];
  if (a == b) return tri_a < tri_b;
  int2 ua = A.ref_up[a], ub = A.ref_up[b];
  while (((uint32_t)ua.x >> 24) > ((uint32_t)ub.x >> 24)) {
    a = ua.x & 0x00ffffff;
    ua = A.ref_up[a];
  }
  while (((uint32_t)ub.x >> 24) > ((uint32_t)ua.x >> 24)) {
    b = ub.x & 0x00ffffff;
    ub = A.ref_up[b];
  }
  while ((ua.x & 0x00ffffff) != (ub.x & 0x00ffffff)) {
    a = ua.x & 0x00ffffff;
    b = ub.x & 0x00ffffff;
    ua = A.ref_up[a];
    ub = A.ref_up[b];
  }
  const float4* r = A.q.sc.inner + (size_t)((uint32_t)ua.y & 0x7fffffffu) * 4;
  const float4 q0 = r[0], q1 = r[1], q2 = r[2];
  const float d1 = hit_aabb(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
  const float d2 = hit_aabb(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
  const bool left_first = d1 < d2; // (both children are hit: both leaves were reached)
  return ((uint32_t)ua.y >> 31) ? !left_first : left_first;
}

// SEMI: rays with an exactly-zero direction component are traversed here (with the NaN watch) instead of sent to the redo
// list.  A template parameter because the watch costs every ray of the launch 2-3 % (a ballot per iteration, four flags,
// registers); the host turns it on for the launches that see such rays in numbers: the MIS integrators' bounce stages.
template <bool REL, bool LOG, int PRUNE, bool GEN, bool SEMI = false, bool GS = false>
EZD void traceq4_body(const TraceQ4Args& A) {
  static_assert(!GEN || REL, "generated rays start at the launch's uniform origin");
  extern __shared__ __attribute__((aligned(16))) int lds_stack[];
  const TraceQArgs& a = A.q;
  // LDS layout: [lane table: BLOCK ints][stack rows][staged records]
  int* stack = lds_stack + BLOCK + threadIdx.x;
  const TraceScene& sc = a.sc;
  const uint32_t n_real = (*a.n_paths) * a.rays_per_path;
  const int lane = threadIdx.x & 63;
  if (n_real == 0) return;
  // (GSCAT: queue positions below are LOGICAL; see TraceQ4Args::gscat_shift.  The logical domain is 256 * R whole granules.)
  // (granule = 8 slots, a compile-time constant, and R is re-derived from n_rays where it is used: the kernel sits at its
  // SGPR limit too, and every uniform value kept across the loop is spilled into a VGPR lane)
  constexpr bool gscat = GS && !GEN;
  constexpr uint32_t GSH = 3u;
  const uint32_t n_rays = gscat ? ((((n_real + (1u << GSH) - 1u) >> GSH) + 255u) >> 8) << (8u + GSH) : n_real;
  int* wsrc = lds_stack + (threadIdx.x >> 6) * 64;
  float4* lds_nodes = reinterpret_cast<float4*>(lds_stack + BLOCK + a.stack_entries * BLOCK);
  const float4* inner = REL ? A.inner4_rel : A.inner4;
  for (int k = threadIdx.x; k < A.lds_nodes4 * 7; k += BLOCK) lds_nodes[k] = inner[(k / 7) * N4_FLOAT4 + (k % 7)];
  if (PRUNE == 2) stack[A.stack_cap * BLOCK] = 0; // (the lane's count of spilled stack entries: TraceQ4Args::stack_cap)
  __syncthreads();

  // queue indices: see ezrt_traceq.h
  const uint32_t n_waves = gridDim.x * (BLOCK / 64);
  uint32_t pool_size = (n_rays + n_waves * a.pool_div - 1) / (n_waves * a.pool_div);
  pool_size = pool_size > a.pool_max ? a.pool_max : (pool_size < a.pool_min ? a.pool_min : pool_size);
  uint32_t static_rounds = (uint32_t)(((unsigned long long)n_rays * a.static_pct) / (100ull * n_waves * pool_size));
  static_rounds = static_rounds < 1u ? 1u : static_rounds;
  const uint32_t static_total = static_rounds * n_waves * pool_size;
  const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  uint32_t round = 0;
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false;

  uint32_t nx_slot = REF_NONE;
  float4 nx_o = make_float4(0, 0, 0, 0), nx_d = make_float4(0, 0, 0, 0);

  uint32_t slot = 0;
  f3 S = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
  float best_t = INF;
  int32_t best_tri = -1;
  int sp = 0, sb = 0;
  bool tie = false;
  bool shared = false;
  int32_t tie_tri = -1; // a second triangle at exactly best_t (ordered against best_tri when the ray is published)
  bool semi = false;    // this lane's ray has a direction component that is exactly +-0 (see "Rays with a zero component")
  bool anyhit = false;  // an env shadow ray: any accepted hit ends the traversal (TraceQArgs::anyhit_even)
  uint32_t ref = REF_NONE;
  uint32_t n_counted = 0;
  unsigned long long* hits64 = reinterpret_cast<unsigned long long*>(a.hits);
  // PRUNE: a slot is skipped when its entry distance exceeds prune_t = (best_t + pdelta)(1 + 2^-19),
  //   pdelta = (prune_a + prune_cs |S|_inf) max_k |inv_k|
  float prune_t = INF, pdelta = 0.0f;
  auto set_delta = [&]() {
    const float sm = REL ? hw_max3(ez_abs(a.origin[0]), ez_abs(a.origin[1]), ez_abs(a.origin[2])) : hw_max3(ez_abs(S.x), ez_abs(S.y), ez_abs(S.z));
    pdelta = (A.prune_a + A.prune_cs * sm) * hw_max3(ez_abs(inv.x), ez_abs(inv.y), ez_abs(inv.z));
    prune_t = (best_t + pdelta) * PRUNE_REL;
  };
  auto to_redo = [&](uint32_t s) { // (rare) the in-order binary kernel decides this ray
    if (atomicExch(&a.redo_flag[s], 1u) == 0u) a.redo_slots[atomicAdd(a.redo_count, 1u)] = s;
    // until it has, the record says so: key (t bits 0, tri HIT_PENDING) is below every real key, so later atomicMin
    // contributions of a split ray cannot replace it; the redo launch overwrites it with a plain store
    atomicMin(&hits64[s], (unsigned long long)(uint32_t)HIT_PENDING);
  };
  constexpr bool RING = PRUNE == 2; // (the slot-order modes keep their exact bound and absolute rows)
  const int ring_mask = RING ? A.stack_cap - 1 : -1; // (-1: p & -1 = p)
  auto finish = [&]() {
    ref = REF_DONE;
    sp = 0;
    sb = 0;
    if (RING) stack[A.stack_cap * BLOCK] = 0; // (nothing spilled: an early end -- any-hit, the redo list -- may leave entries behind)
  };
  // the next pending subtree of this lane's ray, or the end of the ray
  auto pop_or_finish = [&]() {
    if (sp > sb) {
      sp--;
      ref = (uint32_t)stack[(sp & ring_mask) * BLOCK];
    } else if (RING && stack[A.stack_cap * BLOCK] > 0) { // (rare) entries spilled to global memory come back, last in first out
      const int g = stack[A.stack_cap * BLOCK] - 1;
      ref = A.ovf[(size_t)(blockIdx.x * BLOCK + threadIdx.x) * (uint32_t)A.ovf_cap + (uint32_t)g];
      stack[A.stack_cap * BLOCK] = g;
    } else {
      finish();
    }
  };
  auto publish = [&]() {
    if (tie_tri >= 0) { // two candidates at the final distance: the reference keeps the one it finds first
      if (tie_precedes(A, tie_tri, best_tri, REL ? mk(a.origin[0], a.origin[1], a.origin[2]) : S, inv)) best_tri = tie_tri;
      tie_tri = -1;
    }
    if (shared) {
      if (best_tri >= 0) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best_t) << 32) | (uint32_t)best_tri;
        const unsigned long long old = atomicMin(&hits64[slot], key);
        if ((uint32_t)(old >> 32) == (uint32_t)(key >> 32) && (uint32_t)old != (uint32_t)key) tie = true;
      }
    } else {
      a.hits[slot] = make_int2(best_tri, __float_as_int(best_t));
    }
    if (tie) to_redo(slot);
    ref = REF_NONE;
    tie = false;
    shared = false;
  };
  // a closer (or equally close) hit from a leaf: strict < keeps the first found; an exact tie with a
  // different triangle is decided by the reference's visit order, which this kernel does not follow
  auto take = [&](float t, int32_t tri) {
    if (t < best_t) {
      best_t = t;
      best_tri = tri;
      tie_tri = -1; // (a tie at a distance that has just been beaten does not matter any more)
      // (a thief's threshold may come from a hit another contributor of its ray found, which can be closer than this one)
      if (PRUNE) prune_t = hw_min(prune_t, (t + pdelta) * PRUNE_REL);
    } else if (t == best_t && tri != best_tri) {
      if (A.tri_leaf && tie_tri < 0) tie_tri = tri; // ordered against best_tri at publish (tie_precedes)
      else if (tri != tie_tri) tie = true;           // a third candidate (or no tables): the redo list
    }
  };

  const unsigned long long t_start = (LOG && a.wave_log) ? wall_clock64() : 0ull;
  unsigned long long t_exhausted = 0ull; // (debug_stages=2) when this wave first found the queue empty
  uint32_t wave_iters = 0, dbg_inner_lanes = 0, dbg_inner_steps = 0, dbg_leaf_lanes = 0, dbg_leaf_rounds = 0, dbg_busy_lanes = 0,
           dbg_refills = 0, dbg_steals = 0, dbg_max_sp = 0;
  for (;;) {
    if (LOG) wave_iters++;
    // ---- refill (batched: wave-wide code for per-lane events)
    const bool want = ref >= REF_DONE;
    const unsigned long long wantm = ballot(want);
    if (wantm && ((uint32_t)__popcll(wantm) >= a.refill_min || !ballot(ref < REF_DONE))) {
      if (LOG && a.wave_log) dbg_refills++;
      if (ref == REF_DONE) publish();
      if (ref == REF_NONE && nx_slot != REF_NONE) {
        const uint32_t adopted = nx_slot;
        nx_slot = REF_NONE;
        if (nx_d.w != 0.0f) {
          n_counted += a.count_rays;
          slot = adopted;
          // REL (the primary stage): every ray starts at the launch's origin -- a uniform value: no per-lane copy of
          // it, no prefetched origin, none of its shuffles in the leaf and steal phases (7 VGPRs less at a budget of 80)
          S = REL ? mk(a.origin[0], a.origin[1], a.origin[2]) : mk(nx_o.x, nx_o.y, nx_o.z);
          d = mk(nx_d.x, nx_d.y, nx_d.z);
          inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
          best_t = INF;
          best_tri = -1;
          sp = 0;
          sb = 0;
          // a ray that is not tame (NaNs possible in the slab test, monotonicity lost) is not this kernel's: it is
          // "finished" at once with the tie flag set, i.e. published as PENDING and appended to the redo list
          semi = SEMI && ray_is_semi(S, d, inv);
          anyhit = a.anyhit_even != 0u && (adopted & 1u) == 0u;
          tie = !(ray_is_tame(S, inv) || semi) || (a.force_pending && adopted % a.force_pending == 0u);
          tie_tri = -1;
          ref = tie ? REF_DONE : A.root4;
          best_tri = tie ? HIT_PENDING : -1;
          if (PRUNE) set_delta();
        }
      }
      const bool need = nx_slot == REF_NONE && !exhausted;
      const unsigned long long m = ballot(need);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t r = lane_rank(m);
        uint32_t idx;
        bool served;
        if (pool_end - pool_next < cnt) {
          uint32_t base = n_rays;
          if (round < static_rounds) {
            base = (round * n_waves + (wave_id + round * 1223u) % n_waves) * pool_size;
            round++;
          } else if (static_total < n_rays) {
            const uint32_t h = wave_id % TRACE_HEADS;
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(a.head + h * TRACE_HEAD_STRIDE, 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const unsigned long long off = (unsigned long long)(k * TRACE_HEADS + h) * pool_size;
            base = off < (unsigned long long)(n_rays - static_total) ? static_total + (uint32_t)off : n_rays;
          }
          const uint32_t left = pool_end - pool_next;
          uint32_t take_n = cnt - left;
          if (take_n > pool_size) take_n = pool_size;
          idx = (r < left) ? (pool_next + r) : (base + (r - left));
          served = r < left + take_n;
          pool_next = base + take_n;
          pool_end = base + pool_size;
          if (base >= n_rays) exhausted = true;
        } else {
          idx = pool_next + r;
          served = true;
          pool_next += cnt;
          if (pool_next >= n_rays && pool_end >= n_rays) exhausted = true;
        }
        uint32_t rs = idx;
        if (gscat) { // logical -> physical queue position (positions past the queue's end hold no ray)
          const uint32_t g = idx >> GSH;
          rs = (((g & 255u) * (n_rays >> (8u + GSH)) + (g >> 8)) << GSH) | (idx & ((1u << GSH) - 1u));
        }
        if (need && served && idx < n_rays && rs < n_real) {
          if (a.slot_map) rs = a.slot_map[idx];
          nx_slot = rs;
          if (!REL) nx_o = a.const_origin == 1u ? make_float4(a.origin[0], a.origin[1], a.origin[2], 0.0f) : a.rq.o[rs >> (a.const_origin >> 1)];
          if (GEN) { // generate, and leave the direction in the queue for the stage's shading passes (and a redo launch)
            nx_d = primary_dir(A.gen_p, A.gen_blocks, A.gen_div_blocks, A.gen_div_sub, A.gen_scatter, A.gen_scatter_shift, A.gen_frame_first, rs);
            a.rq.d[rs] = nx_d;
          } else {
            nx_d = a.rq.d[rs];
          }
        }
      }
    }
    if (!ballot((ref & nx_slot) != REF_NONE)) {
      // nobody holds or has prefetched a ray.  That only ends the wave once it has nothing left to DRAW: under the
      // scattered draw a whole pool can fall into the padding past the queue's end (rs >= n_real) while later static
      // rounds still hold rays (pool_max 8 with static_pct 50 left 96 of 400 000 rays untraced)
      if (!exhausted) continue;
      break;
    }
    if (LOG && a.wave_log && exhausted && !t_exhausted) t_exhausted = wall_clock64();

    // ---- work stealing: lanes with nothing left to fetch take the oldest pending subtree of a busy lane
    if (a.steal && exhausted) {
      if (A.handover) { // idle lanes take the prefetched rays of lanes that are still traversing (see TraceQ4Args::handover)
        const bool idle0 = (ref & nx_slot) == REF_NONE;
        const bool host = nx_slot != REF_NONE && ref < REF_DONE;
        const unsigned long long im0 = ballot(idle0), hm = ballot(host);
        if (im0 && hm) {
          const int ni0 = (int)__popcll(im0), nh = (int)__popcll(hm);
          const int n0 = ni0 < nh ? ni0 : nh;
          const int ir0 = (int)lane_rank(im0), hr = (int)lane_rank(hm);
          const bool giver = host && hr < n0, taker = idle0 && ir0 < n0;
          if (giver) wsrc[hr] = lane;
          __builtin_amdgcn_wave_barrier();
          const int src = taker ? wsrc[ir0] : lane;
          const uint32_t hs = (uint32_t)__shfl((int)nx_slot, src, 64);
          const float hdx = __shfl(nx_d.x, src, 64), hdy = __shfl(nx_d.y, src, 64), hdz = __shfl(nx_d.z, src, 64), hdw = __shfl(nx_d.w, src, 64);
          float hox = 0.0f, hoy = 0.0f, hoz = 0.0f;
          if (!REL) hox = __shfl(nx_o.x, src, 64), hoy = __shfl(nx_o.y, src, 64), hoz = __shfl(nx_o.z, src, 64);
          __builtin_amdgcn_wave_barrier();
          if (giver) nx_slot = REF_NONE;
          if (taker) { // (adopted by the next refill, like a ray this lane had prefetched itself)
            nx_slot = hs;
            nx_d = make_float4(hdx, hdy, hdz, hdw);
            if (!REL) nx_o = make_float4(hox, hoy, hoz, 0.0f);
          }
        }
      }
      const bool idle = (ref & nx_slot) == REF_NONE;
      const unsigned long long im = ballot(idle);
      if (im) {
        const bool rich = sp > sb;
        const unsigned long long vm = ballot(rich);
        const int ni = (int)__popcll(im), nv = (int)__popcll(vm);
        const int n = ni < nv ? ni : nv;
        const int ir = (int)lane_rank(im), vr = (int)lane_rank(vm);
        if (n) {
          if (LOG && a.wave_log) dbg_steals++;
          const bool victim = rich && vr < n, thief = idle && ir < n;
          int give = 0;
          if (victim) {
            wsrc[vr] = lane;
            give = stack[(sb & ring_mask) * BLOCK];
            sb++;
            if (!shared) atomicExch(&hits64[slot], ~0ull); // first split: "no hit yet" (see ezrt_traceq.h)
            shared = true;
          }
          __builtin_amdgcn_wave_barrier();
          const int src = thief ? wsrc[ir] : lane;
          const int got = __shfl(give, src, 64);
          const uint32_t vslot = (uint32_t)__shfl((int)slot, src, 64);
          const float vsx = REL ? a.origin[0] : __shfl(S.x, src, 64), vsy = REL ? a.origin[1] : __shfl(S.y, src, 64),
                      vsz = REL ? a.origin[2] : __shfl(S.z, src, 64);
          const float vdx = __shfl(d.x, src, 64), vdy = __shfl(d.y, src, 64), vdz = __shfl(d.z, src, 64);
          const int vsemi = __shfl((int)semi, src, 64), vany = __shfl((int)anyhit, src, 64);
          // the victim's best hit so far bounds the thief's pruning too (knob steal_bound; round 5): a real hit of the SAME ray, so the
          // final minimum is no larger and the proven margin applies -- the oldest pending row is the farthest subtree under the
          // nearest-first order, the one most likely to lie wholly behind that hit, and the thief used to walk it with best_t = INF
          const float vbt = (PRUNE && A.steal_bound) ? __shfl(best_t, src, 64) : INF;
          if (thief) {
            semi = vsemi != 0;
            anyhit = vany != 0;
            shared = true;
            slot = vslot;
            S = mk(vsx, vsy, vsz);
            d = mk(vdx, vdy, vdz);
            inv = mk(ez_rcp(vdx), ez_rcp(vdy), ez_rcp(vdz));
            best_t = INF;
            best_tri = -1;
            sp = 0;
            sb = 0;
            ref = (uint32_t)got;
            if (PRUNE) set_delta();
            if (PRUNE) prune_t = hw_min(prune_t, (vbt + pdelta) * PRUNE_REL);
          }
        }
      }
    }

    // ---- inner step: four slab tests (hitAABB, P5/fsh:220-233) on one 4-wide record
    const bool at_inner = (int32_t)ref >= 0;
    if (LOG && a.wave_log) {
      const uint32_t ni = (uint32_t)__popcll(ballot(at_inner));
      dbg_inner_lanes += ni;
      dbg_inner_steps += ni ? 1u : 0u;
      dbg_busy_lanes += (uint32_t)__popcll(ballot(ref < REF_DONE));
    }
    if (at_inner) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) const v4f lds_v4f;
      typedef __attribute__((address_space(3))) const char lds_char;
      // byte offset of the NEAR row of each axis inside a record: 0 (AA) for inv >= 0, 64 (BB) for inv < 0
      const uint32_t kx = (__float_as_uint(inv.x) >> 31) << 6, ky = (__float_as_uint(inv.y) >> 31) << 6,
                     kz = (__float_as_uint(inv.z) >> 31) << 6;
      v4f nx, ny, nz, fx, fy, fz, rfv;
      // PRUNE: a reference may carry REF_NOPRUNE (nothing below this record is ever skipped)
      const uint32_t rec = ref & REF_INDEX; // (also when PRUNE == 0: the flag is part of the scene's records, the mode is a knob)
      const float thr = (PRUNE && ref >= REF_NOPRUNE) ? __builtin_inff() : prune_t;
      if (rec < (uint32_t)A.lds_nodes4) { // top of the tree: staged in LDS
        lds_char* lb = (lds_char*)lds_nodes + rec * (uint32_t)(N4_LDS_DWORDS * 4);
        nx = *(lds_v4f*)(lb + kx);
        ny = *(lds_v4f*)(lb + 16u + ky);
        nz = *(lds_v4f*)(lb + 32u + kz);
        fx = *(lds_v4f*)(lb + 64u - kx);
        fy = *(lds_v4f*)(lb + 80u - ky);
        fz = *(lds_v4f*)(lb + 96u - kz);
        rfv = *(lds_v4f*)(lb + 48u);
      } else {
        // (32-bit offsets from the uniform table pointer: global_load with an SGPR base and one VGPR offset each,
        // no 64-bit address arithmetic; the table is < 2^32 bytes: n_inner4 < 2^24 records of 128 B)
        const char* tab = reinterpret_cast<const char*>(inner);
        const uint32_t roff = rec * (uint32_t)(N4_FLOAT4 * 16);
        nx = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + kx));
        ny = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + ky + 16u));
        nz = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + kz + 32u));
        fx = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + 64u - kx));
        fy = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + 80u - ky));
        fz = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + 96u - kz));
        rfv = *reinterpret_cast<const v4f*>(tab + (uint32_t)(roff + 48u));
      }
      const float4 rf = make_float4(rfv.x, rfv.y, rfv.z, rfv.w);
      float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, e3 = 0.0f; // PRUNE == 2: entry distances of the four slots
      // some lane of the wave holds a ray with a zero direction component: its slab products can be NaN (0 x inf), which
      // the hardware min3 / max3 would silently drop -- such a lane leaves for the redo list the moment one turns up
      const bool wave_semi = SEMI && ballot(semi && at_inner) != 0ull;
      bool n0 = false, n1 = false, n2 = false, n3 = false; // a NaN among the slot's six products
      auto slab = [&](float nxk, float nyk, float nzk, float fxk, float fyk, float fzk, float& entry, bool& nan_seen) -> bool {
        float t0x, t0y, t0z, t1x, t1y, t1z;
        if (REL) { // boxes already translated by the common origin
          t0x = nxk * inv.x, t0y = nyk * inv.y, t0z = nzk * inv.z;
          t1x = fxk * inv.x, t1y = fyk * inv.y, t1z = fzk * inv.z;
        } else {
          t0x = (nxk - S.x) * inv.x, t0y = (nyk - S.y) * inv.y, t0z = (nzk - S.z) * inv.z;
          t1x = (fxk - S.x) * inv.x, t1y = (fyk - S.y) * inv.y, t1z = (fzk - S.z) * inv.z;
        }
        const float t1 = hw_min3(t1x, t1y, t1z);
        const float t0 = hw_max3(t0x, t0y, t0z);
        if (wave_semi) nan_seen = ((t0x != t0x) | (t0y != t0y) | (t0z != t0z) | (t1x != t1x) | (t1y != t1y) | (t1z != t1z)) != 0;
        if (PRUNE == 2) entry = t0;
        // == hitAABB(..) > 0; PRUNE: ... and not provably beyond the best hit so far (see "Distance pruning")
        if (PRUNE) return (t1 >= t0) && (t1 > 0.0f) && !(t0 > thr);
        return (t1 >= t0) && (t1 > 0.0f);
      };
      const bool h0 = slab(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, e0, n0);
      const bool h1 = slab(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, e1, n1);
      const bool h2 = slab(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, e2, n2);
      const bool h3 = slab(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, e3, n3);
      const uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z),
                     r3 = __float_as_uint(rf.w);
      // (unused slots hold NaN boxes on purpose: only a NaN in a USED slot means 0 x inf)
      const bool nan_slot = wave_semi && semi && ((n0 && r0 != REF_EMPTY) || (n1 && r1 != REF_EMPTY) || (n2 && r2 != REF_EMPTY) || (n3 && r3 != REF_EMPTY));
      if (nan_slot) { // (rare) 0 x inf in a slab product: the NaN-aware in-order kernel decides this ray
        tie = true;
        finish();
      } else if (PRUNE == 2) {
        // nearest first: continue with the hit slot of the smallest entry distance (so that the first leaves a ray
        // reaches are the likely winners and the rest gets pruned), push the other hit slots highest-first.  The order
        // is a heuristic only: pruning is exact whatever the order, exact ties still go to the redo list.
        const float inf = __builtin_inff();
        const float k0 = h0 ? e0 : inf, k1 = h1 ? e1 : inf, k2 = h2 ? e2 : inf, k3 = h3 ? e3 : inf;
        const float km = hw_min(hw_min3(k0, k1, k2), k3);
        const bool s0 = h0 && k0 == km;
        const bool s1 = !s0 && h1 && k1 == km;
        const bool s2 = !s0 && !s1 && h2 && k2 == km;
        const bool s3 = !s0 && !s1 && !s2 && h3;
        stack[(sp & ring_mask) * BLOCK] = (int)r3; // (written unconditionally -- a row above the top is free: see the eviction threshold -- and
        sp += (h3 && !s3) ? 1 : 0;                    // kept only when the slot is hit and not the one continued with: no divergent branch per push)
        stack[(sp & ring_mask) * BLOCK] = (int)r2; // (written unconditionally -- a row above the top is free: see the eviction threshold -- and
        sp += (h2 && !s2) ? 1 : 0;                    // kept only when the slot is hit and not the one continued with: no divergent branch per push)
        stack[(sp & ring_mask) * BLOCK] = (int)r1; // (written unconditionally -- a row above the top is free: see the eviction threshold -- and
        sp += (h1 && !s1) ? 1 : 0;                    // kept only when the slot is hit and not the one continued with: no divergent branch per push)
        stack[(sp & ring_mask) * BLOCK] = (int)r0; // (written unconditionally -- a row above the top is free: see the eviction threshold -- and
        sp += (h0 && !s0) ? 1 : 0;                    // kept only when the slot is hit and not the one continued with: no divergent branch per push)
        if (h0 || h1 || h2 || h3) {
          ref = s0 ? r0 : (s1 ? r1 : (s2 ? r2 : r3));
          // this order has no small worst-case stack bound (up to three pending entries per level): when the ring has fewer than
          // three free rows left the oldest four entries are spilled (TraceQ4Args::stack_cap); a lane whose spill area is full as
          // well hands its ray to the redo list (in-order binary kernel, any depth <= 63)
          if (sp - sb > A.stack_cap - 4) { // (four rows free for the next step: three pushes and the unconditional write above them)
            const int g = stack[A.stack_cap * BLOCK];
            const int live = sp - sb, n_ev = live < 4 ? live : 4; // (live >= 1 here; a ring of 4 rows -- the test hook -- keeps none)
            if (g + n_ev > A.ovf_cap) {
              tie = true;
              finish();
            } else {
              uint32_t* o = A.ovf + (size_t)(blockIdx.x * BLOCK + threadIdx.x) * (uint32_t)A.ovf_cap + (uint32_t)g;
              for (int e = 0; e < n_ev; e++) o[e] = (uint32_t)stack[((sb + e) & ring_mask) * BLOCK];
              sb += n_ev;
              stack[A.stack_cap * BLOCK] = g + n_ev;
            }
          }
        } else {
          pop_or_finish();
        }
      } else {
      // visit the hit slots in ascending order: continue with the lowest, push the others highest-first
      if (h3 && (h0 || h1 || h2)) {
        stack[sp * BLOCK] = (int)r3;
        sp++;
      }
      if (h2 && (h0 || h1)) {
        stack[sp * BLOCK] = (int)r2;
        sp++;
      }
      if (h1 && h0) {
        stack[sp * BLOCK] = (int)r1;
        sp++;
      }
      if (h0 || h1 || h2 || h3) {
        ref = h0 ? r0 : (h1 ? r1 : (h2 ? r2 : r3));
      } else {
        pop_or_finish();
      }
      }
    }

    if (LOG && a.wave_log) dbg_max_sp = dbg_max_sp > (uint32_t)sp ? dbg_max_sp : (uint32_t)sp;
    // ---- leaf phase (hitArray, P5/fsh:238-251): postponed until enough lanes wait at a leaf, or nobody can step
    const bool at_leaf = (int32_t)ref < -3; // bit 31 set, not REF_NONE / REF_DONE / REF_EMPTY
    const unsigned long long lm = ballot(at_leaf);
    if (lm) {
      const int Lc = (int)__popcll(lm);
      const bool go = Lc >= a.leaf_threshold || !ballot((int32_t)ref >= 0);
      if (go) {
        if (Lc <= 32) {
          // cooperative: g = 64 / Lc lanes (power of two) per waiting ray
          const int sh = (Lc <= 1) ? 0 : (32 - __clz(Lc - 1));
          const int g = 64 >> sh;
          const uint32_t rank = lane_rank(lm);
          if (at_leaf) wsrc[rank] = lane;
          __builtin_amdgcn_wave_barrier();
          const int grp = lane >> (6 - sh), m = lane & (g - 1);
          const bool helper = grp < Lc;
          if (LOG && a.wave_log) {
            dbg_leaf_lanes += (uint32_t)Lc;
            dbg_leaf_rounds++;
          }
          const int src = helper ? wsrc[grp] : lane;
          const f3 cS = REL ? mk(a.origin[0], a.origin[1], a.origin[2])
                            : mk(__shfl(S.x, src, 64), __shfl(S.y, src, 64), __shfl(S.z, src, 64));
          const f3 cd = mk(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
          const uint32_t lref = (uint32_t)__shfl((int)ref, src, 64);
          unsigned long long key = ~0ull;
          if (helper) {
            const int first = (int)(lref & 0x00ffffffu);
            const int n = (int)((lref >> 24) & 0x7fu) + 1;
            for (int k = m; k < n; k += g) {
              float t;
              if (hit_triangle_t(sc.tri_geom + (size_t)(first + k) * 3, cS, cd, t)) {
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)(first + k);
                key = k2 < key ? k2 : key;
              }
            }
          }
          for (int off = 1; off < g; off <<= 1) {
            const unsigned long long other = __shfl_xor(key, off, 64);
            key = other < key ? other : key;
          }
          const unsigned long long mine = __shfl(key, (int)(rank << (6 - sh)), 64);
          if (at_leaf && mine != ~0ull) take(__uint_as_float((uint32_t)(mine >> 32)), (int32_t)(uint32_t)mine);
        } else if (at_leaf) {
          const int first = (int)(ref & 0x00ffffffu);
          const int n = (int)((ref >> 24) & 0x7fu) + 1;
          for (int i = first; i < first + n; i++) {
            float t;
            if (hit_triangle_t(sc.tri_geom + (size_t)i * 3, REL ? mk(a.origin[0], a.origin[1], a.origin[2]) : S, d, t)) take(t, i);
          }
        }
        if (at_leaf) {
          if (anyhit && best_tri >= 0) finish(); // (an env shadow ray that has hit something is done: only isHit is asked of it)
          else pop_or_finish();
        }
      }
    }
  }

  const unsigned long long rr = wave_sum(n_counted);
  if (lane == 0 && rr) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_RAYS], rr);
  if (LOG && a.wave_log && lane == 0) {
    unsigned long long* w = a.wave_log + (size_t)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 8;
    w[0] = t_start;
    w[1] = wall_clock64();
    w[2] = wave_iters | ((unsigned long long)dbg_inner_steps << 32);
    w[3] = rr | ((unsigned long long)dbg_inner_lanes << 32);
    w[4] = dbg_leaf_lanes | ((unsigned long long)dbg_leaf_rounds << 32);
    unsigned long long mx = dbg_max_sp; // the highest stack row any lane of the wave reached
    for (int off = 32; off; off >>= 1) {
      const unsigned long long o = __shfl_xor(mx, off, 64);
      mx = o > mx ? o : mx;
    }
    w[5] = dbg_busy_lanes | (mx << 48);
    w[6] = dbg_refills | ((unsigned long long)dbg_steals << 32);
    w[7] = t_exhausted;
  }
}

// GS: queue positions are drawn in the scattered order (TraceQ4Args::gscat_shift; granules of 8 slots)
template <int WPS, bool REL, bool LOG = false, int PRUNE = 0, bool GEN = false, bool SEMI = false, bool GS = false>
__global__ __launch_bounds__(BLOCK, WPS) void traceq4_kernel(TraceQ4Args A) {
  traceq4_body<REL, LOG, PRUNE, GEN, SEMI, GS>(A);
}

} // namespace ezd
