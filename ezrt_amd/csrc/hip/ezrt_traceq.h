// ezrt_traceq.h -- traceq_kernel: persistent hitBVH over a ray queue (the dominant kernel).
//
// Semantics per ray = hitBVH + hitArray + hitTriangle + hitAABB of the reference
// (P5/fsh:160-306): unpruned, near child first, ties right-first, strict < keeps the first-found
// hit.  Only the *schedule* is MI355X-specific:
//
//  * persistent workgroups; each lane owns one ray at a time and keeps its NEXT ray prefetched in
//    registers, so a lane that finishes refills without a memory round trip and a wave never
//    idles behind its slowest ray.  Waves own interleaved static pools of queue indices and
//    reserve the rest dynamically (one atomic per pool of 256);
//  * per-lane events are batched: storing a finished ray's result, adopting the prefetched ray and
//    prefetching the next one is wave-wide code, so it only runs once `refill_min` lanes are free;
//  * lane state lives in `ref` (node / leaf / REF_DONE / REF_NONE): every predicate is one compare;
//  * the breadth-first top of the tree is staged in LDS once per workgroup (80-B stride, explicit
//    LDS address space so the compiler emits ds_read_b128, not flat loads);
//  * per-lane traversal stack in LDS, stack[row][256]: bank = lane % 32, conflict-free;
//  * leaves are postponed until >= leaf_threshold lanes wait at one (ballot) or nobody can step;
//    then the waiting rays share the WHOLE wave: g = 64 / Lc lanes per ray test triangles m, m+g,
//    .. and a 64-bit (t bits, index) minimum across the group reproduces hitArray's first-minimum;
//  * slab tests use v_min/v_max/v_min3/v_max3 for rays whose origin and 1/direction are finite
//    (no NaN can arise; see hit_aabb_tame), the exact select form otherwise.
#pragma once
#include "ezrt_kernels.h"

namespace ezd {

constexpr uint32_t REF_NONE = 0xffffffffu; // a lane without a ray (every predicate below is ONE compare on `ref`)
constexpr uint32_t REF_DONE = 0xfffffffeu; // traversal finished, {t, triangle} not stored yet (done in the batched refill)
#ifndef EZRT_TRACE_HEADS
#define EZRT_TRACE_HEADS 8
#endif
constexpr uint32_t TRACE_HEADS = EZRT_TRACE_HEADS; // reservation counters per queue (-DEZRT_TRACE_HEADS=n: A/B builds, tools/build_variant.sh)
constexpr uint32_t TRACE_HEAD_STRIDE = 16; // words between them (64 bytes)
constexpr uint32_t TRACE_POOL_MIN = 8; // smallest reservation: short queues are spread over every wave

struct RayQueue {
  float4* o; // origin.xyz, -   (not stored for primary rays: TraceQArgs.origin)
  float4* d; // dir.xyz, valid (1) / skip (0)
};

// what hitBVH needs of the scene
struct TraceScene {
  const float4* tri_geom;
  const float4* inner;
  uint32_t root_ref;
};
inline TraceScene trace_scene(const DevScene& d) { return TraceScene{d.tri_geom, d.inner, d.root_ref}; }

struct TraceQArgs {
  TraceScene sc; // (not the whole DevScene: every kernel argument a launch carries costs the trace kernels scalar
                 // registers, and they sit at their VGPR line -- two more pointers in DevScene turned 4 spills into 7)
  RayQueue rq;
  int2* hits;
  const uint32_t* n_paths; // device count; rays = n_paths * rays_per_path
  uint32_t rays_per_path;
  uint32_t const_origin;   // 1: every ray of this queue starts at `origin` (primary rays); rq.o is not read.  2: the two rays of a
                           // path (MIS: shadow ray, bounce ray) share one stored origin, rq.o[slot >> 1].  0: rq.o[slot]
  float origin[3];
  const float4* inner_rel; // const_origin only (or NULL): sc.inner with every box already translated by -origin, i.e.
                           // (AA - S, BB - S) evaluated once per record instead of once per visit -- the same fp32
                           // subtractions, so the same bits
  uint32_t* head;          // TRACE_HEADS counters (TRACE_HEAD_STRIDE words apart) of the dynamically reserved part of
                           // the queue (device, zeroed per launch)
  unsigned long long* counters;
  int32_t leaf_threshold;  // lanes waiting at a leaf that trigger the triangle phase
  uint32_t refill_min;     // lanes that must be free before the wave runs its refill code
  uint32_t static_pct;     // share of the queue dealt statically (interleaved rounds of pools), percent
  uint32_t pool_div, pool_max; // pool = clamp(n_rays / (n_waves * pool_div), pool_min, pool_max)
  uint32_t pool_min;           // >= TRACE_POOL_MIN: a short queue is dealt to fewer waves, this many rays each (the others retire at once)
  int32_t lds_nodes;       // inner records [0, lds_nodes) staged in LDS (after stack + lane table)
  int32_t stack_entries;   // LDS stack rows (tree depth); the per-wave lane table follows them
  uint32_t* dbg;           // diagnostic (FULLCTR only): [0] max pops/ray [1] max tris/ray [2] max iterations/ray
  const uint32_t* slot_map; // optional indirection: queue index -> ray slot (a redo list)
  // Work stealing (steal != 0): a lane with nothing left to fetch takes the OLDEST pending subtree of a
  // busy lane of its wave and traverses it for that lane's ray.  hitBVH is unpruned, so subtrees are
  // independent and the ray's result is the minimum over all of them: every contributor merges its
  // {t, triangle} into hits[slot] with one 64-bit atomicMin (the record is reset to ~0 = "no hit yet"
  // when the ray is first split; rays that are never split store their result plainly).  Order then
  // only matters for EXACT ties in t; whoever sees one appends the ray to the redo list, which a
  // second launch (steal = 0, reference order, plain stores) re-traces.  count_rays = 0 in that launch.
  uint32_t steal;
  uint32_t count_rays;
  uint32_t* redo_count;
  uint32_t* redo_slots;
  uint32_t* redo_flag; // one word per ray slot: a ray is appended once (cleared again by the redo launch)
  uint32_t force_pending;       // test hook (knob debug_force_pending = k > 0): traceq4_kernel treats every ray whose slot is a
                                // multiple of k as not tame, i.e. sends it through the pending -> redo -> second-pass route
  // Distance pruning in THIS kernel's in-order traversal (prune_on != 0, never with FULLCTR: the counters are the unpruned
  // reference's).  The margin and its proof are ezrt_traceq4.h's ("Distance pruning"); here the visit ORDER stays the
  // reference's too -- a pruned child holds no triangle with t <= best_t, so neither the winner nor the first-found
  // among exact ties changes -- which is what the redo launches (ties of the 4-wide kernel, rays that are not tame) need:
  // without it a handful of tied rays walked the whole unpruned tree in ONE lane each (C5: 0.7 ms per stage, 11 % of a frame).
  // A record's third word pair flags children with a triangle below them that has no useful bound.
  uint32_t prune_on;
  float prune_a, prune_cs;
  // 1 (traceq4_kernel, queues with two rays per path): the even slots are env shadow rays, of which the shading stage only
  // asks WHETHER they hit anything (P5/fsh:826-829 `if(!hdrHit.isHit)`): their traversal stops at the first accepted hit.
  // The record then holds A hit, not the closest one -- never set for the audit routes, which report {triangle, t}.
  uint32_t anyhit_even;
  unsigned long long* wave_log; // diagnostic (debug_stages=2): per wave 8 words {start, end (100 MHz ticks), iterations |
                                // inner steps, rays | inner lanes, leaf rays | leaf rounds, busy lanes, -, -}
};

// Hit record (tri) of a ray traceq4_kernel handed to the redo list WITHOUT an answer (a ray that is not tame): the
// shading stage's first pass, which runs while the redo launch is still tracing, defers such a path to its second
// pass instead of reading it as a miss.  (A ray on the redo list because of an exact tie already has a hit record --
// one of the tied triangles -- and "hit" is all the first pass needs to know.)
constexpr int32_t HIT_PENDING = -5;

EZD uint32_t lane_rank(unsigned long long mask) { // number of set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool FULLCTR, int WPS>
__global__ __launch_bounds__(BLOCK, WPS) void traceq_kernel(TraceQArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_stack[];
  int* stack = lds_stack + threadIdx.x;
  const TraceScene& sc = a.sc;
  const uint32_t n_rays = (*a.n_paths) * a.rays_per_path;
  const int lane = threadIdx.x & 63;
  if (n_rays == 0) return; // (redo launches are normally empty)
  int* wsrc = lds_stack + a.stack_entries * BLOCK + (threadIdx.x >> 6) * 64;
  float4* lds_nodes = reinterpret_cast<float4*>(lds_stack + a.stack_entries * BLOCK + BLOCK);
  const bool rel = a.inner_rel != nullptr;
  const float4* inner = rel ? a.inner_rel : sc.inner;
  for (int k = threadIdx.x; k < a.lds_nodes * 4; k += BLOCK) lds_nodes[(k >> 2) * 5 + (k & 3)] = inner[k];
  __syncthreads();

  const uint32_t n_waves = gridDim.x * (BLOCK / 64);
  // Queue indices are handed out in pools of `pool_size`.  The first `static_rounds` pools of every
  // wave are static and interleaved (round k: wave w owns pool k * n_waves + w): no atomic at all for
  // queues of up to n_waves * pool_max rays -- 5120 waves bumping one counter would cost ~60 us per
  // round, the whole budget of a late bounce (one word sustains ~88 atomics/us chip-wide) -- and,
  // with the blocks of a frame queued most-expensive-first, every wave gets the same mix of expensive
  // and cheap pools.  Indices beyond the static region are reserved dynamically, one atomic per pool, on eight counters.
  // (Smaller pools only for the queue's last stretch, and guided pool sizes, were tried: no gain.)
  uint32_t pool_size = (n_rays + n_waves * a.pool_div - 1) / (n_waves * a.pool_div);
  pool_size = pool_size > a.pool_max ? a.pool_max : (pool_size < a.pool_min ? a.pool_min : pool_size);
  uint32_t static_rounds = (uint32_t)(((unsigned long long)n_rays * a.static_pct) / (100ull * n_waves * pool_size));
  static_rounds = static_rounds < 1u ? 1u : static_rounds;
  const uint32_t static_total = static_rounds * n_waves * pool_size;
  const uint32_t wave_id = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  uint32_t round = 0;
  uint32_t pool_next = 0, pool_end = 0; // wave-uniform pool of ray indices
  bool exhausted = false;               // wave-uniform: nothing left to reserve

  uint32_t nx_slot = REF_NONE; // prefetched next ray of this lane (REF_NONE: none)
  float4 nx_o = make_float4(0, 0, 0, 0), nx_d = make_float4(0, 0, 0, 0);

  bool wild = false; // ... needs the exact NaN-aware slab test
  bool wave_wild = false; // wave-uniform: some lane may hold such a ray (refreshed whenever rays are adopted or stolen)
  uint32_t slot = 0;
  f3 S = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
  float best_t = INF;
  int32_t best_tri = -1;
  int sp = 0, sb = 0; // live stack entries are rows [sb, sp): the owner pops at sp, thieves take row sb
  bool tie = false;    // two contributors published the same best distance for different triangles
  bool shared = false; // this lane's ray has been split: other lanes hold subtrees of it (or this lane is a thief)
  uint32_t ref = REF_NONE; // current node: inner record index (bit 31 clear), leaf ref (bit 31 set), or REF_NONE
  const bool pruning = !FULLCTR && a.prune_on != 0u; // (wave-uniform)
  float prune_t = INF, pdelta = 0.0f;                // skip a child whose entry distance exceeds (best_t + pdelta)(1 + 2^-19)
  constexpr float PRUNE_REL_B = 1.0f + 1.0f / 524288.0f;
  auto set_delta = [&]() {
    pdelta = (a.prune_a + a.prune_cs * hw_max3(ez_abs(S.x), ez_abs(S.y), ez_abs(S.z))) * hw_max3(ez_abs(inv.x), ez_abs(inv.y), ez_abs(inv.z));
    prune_t = (best_t + pdelta) * PRUNE_REL_B;
  };
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t ray_p0 = 0, ray_t0 = 0, ray_i0 = 0, iters = 0;
  unsigned long long* hits64 = reinterpret_cast<unsigned long long*>(a.hits);

  // end of this lane's (sub)traversal; the result is stored by publish() when the wave next runs its refill code
  auto finish = [&]() {
    ref = REF_DONE;
    sp = 0;
    sb = 0;
  };
  auto publish = [&]() {
    if (shared) { // several lanes contribute to this ray: merge with a 64-bit atomicMin
      if (best_tri >= 0) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best_t) << 32) | (uint32_t)best_tri;
        const unsigned long long old = atomicMin(&hits64[slot], key);
        if ((uint32_t)(old >> 32) == (uint32_t)(key >> 32) && (uint32_t)old != (uint32_t)key) tie = true;
      }
      if (tie && atomicExch(&a.redo_flag[slot], 1u) == 0u) a.redo_slots[atomicAdd(a.redo_count, 1u)] = slot; // rare
    } else {
      a.hits[slot] = make_int2(best_tri, __float_as_int(best_t));
    }
    ref = REF_NONE;
    tie = false;
    shared = false;
    if (FULLCTR && a.dbg) {
      atomicMax(a.dbg, ctr.pops - ray_p0);
      atomicMax(a.dbg + 1, ctr.tris - ray_t0);
      atomicMax(a.dbg + 2, iters - ray_i0);
    }
  };

  const unsigned long long t_start = a.wave_log ? wall_clock64() : 0ull;
  uint32_t wave_iters = 0, dbg_inner_lanes = 0, dbg_inner_steps = 0, dbg_leaf_lanes = 0, dbg_leaf_rounds = 0, dbg_busy_lanes = 0;
  for (;;) {
    if (FULLCTR) iters++;
    wave_iters++;
    // ---- refill: lanes without work adopt their prefetched ray, then prefetch another
    const bool want = ref >= REF_DONE;
    const unsigned long long wantm = ballot(want);
    // (the refill code runs for the whole wave: batch it until a few lanes are free, or nobody has a ray)
    if (wantm && ((uint32_t)__popcll(wantm) >= a.refill_min || !ballot(ref < REF_DONE))) {
      if (ref == REF_DONE) publish();
      if (want && nx_slot != REF_NONE) {
        const uint32_t adopted = nx_slot;
        nx_slot = REF_NONE;
        if (nx_d.w != 0.0f) {
          slot = adopted;
          S = mk(nx_o.x, nx_o.y, nx_o.z);
          d = mk(nx_d.x, nx_d.y, nx_d.z);
          inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
          wild = !ray_is_tame(S, inv);
          best_t = INF;
          best_tri = -1;
          sp = 0;
          sb = 0;
          ref = sc.root_ref;
          if (pruning) set_delta();
          ctr.rays += a.count_rays;
          if (FULLCTR) {
            ray_p0 = ctr.pops;
            ray_t0 = ctr.tris;
            ray_i0 = iters;
            ctr.pops++;
          }
        }
      }
      wave_wild = ballot(wild && ref < REF_DONE) != 0ull;
      const bool need = nx_slot == REF_NONE && !exhausted;
      const unsigned long long m = ballot(need);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t r = lane_rank(m);
        uint32_t idx;
        bool served;
        if (pool_end - pool_next < cnt) { // wave-uniform: top the pool up with one atomic
          uint32_t base = n_rays; // (nothing beyond the static region: done)
          if (round < static_rounds) {
            // (rotated per round: n_waves is often a multiple of the blocks per frame, and a wave must not
            // meet the same block of every frame)
            base = (round * n_waves + (wave_id + round * 1223u) % n_waves) * pool_size;
            round++;
          } else if (static_total < n_rays) {
            // dynamic pool p belongs to counter p % TRACE_HEADS; a wave only ever asks its own counter (every
            // counter has ~n_waves / TRACE_HEADS clients that keep asking until it runs dry, so no pool is
            // left behind): the reservations of a stage spread over eight words instead of queueing on one
            const uint32_t h = wave_id % TRACE_HEADS;
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(a.head + h * TRACE_HEAD_STRIDE, 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const unsigned long long off = (unsigned long long)(k * TRACE_HEADS + h) * pool_size;
            base = off < (unsigned long long)(n_rays - static_total) ? static_total + (uint32_t)off : n_rays;
          }
          const uint32_t left = pool_end - pool_next; // hand out the old pool's rest first
          uint32_t take = cnt - left;
          if (take > pool_size) take = pool_size;
          idx = (r < left) ? (pool_next + r) : (base + (r - left));
          served = r < left + take;
          pool_next = base + take;
          pool_end = base + pool_size;
          if (base >= n_rays) exhausted = true;
        } else {
          idx = pool_next + r;
          served = true;
          pool_next += cnt;
          if (pool_next >= n_rays && pool_end >= n_rays) exhausted = true;
        }
        if (need && served && idx < n_rays) {
          uint32_t rs = idx;
          if (a.slot_map) {
            rs = a.slot_map[idx];
            if (a.redo_flag) a.redo_flag[rs] = 0u;
          }
          nx_slot = rs;
          nx_o = a.const_origin == 1u ? make_float4(a.origin[0], a.origin[1], a.origin[2], 0.0f) : a.rq.o[rs >> (a.const_origin >> 1)];
          nx_d = a.rq.d[rs];
        }
      }
    }
    if (!ballot((ref & nx_slot) != REF_NONE)) break; // no lane has a ray or a prefetched one

    // ---- work stealing: lanes with nothing left to fetch take the oldest pending subtree of a busy lane
    if (a.steal && exhausted) { // (while the queue still has rays, free lanes are about to be refilled)
      const bool idle = (ref & nx_slot) == REF_NONE;
      const unsigned long long im = ballot(idle);
      if (im) {
        const bool rich = sp > sb; // (sp == sb == 0 without a ray)
        const unsigned long long vm = ballot(rich);
        if (vm) {
          const int ni = (int)__popcll(im), nv = (int)__popcll(vm);
          const int n = ni < nv ? ni : nv;
          const int ir = (int)lane_rank(im), vr = (int)lane_rank(vm);
          const bool victim = rich && vr < n, thief = idle && ir < n;
          int give = 0;
          if (victim) { // hand over the bottom row
            wsrc[vr] = lane;
            give = stack[sb * BLOCK];
            sb++;
            // first split of this ray: reset its record to "no hit yet" (~0).  The thieves' atomicMins
            // are issued by this same wave, later in program order, to the same address: they reach
            // the L2 after this atomic.
            if (!shared) atomicExch(&hits64[slot], ~0ull);
            shared = true;
          }
          __builtin_amdgcn_wave_barrier();
          const int src = thief ? wsrc[ir] : lane;
          const int got = __shfl(give, src, 64);
          const uint32_t vslot = (uint32_t)__shfl((int)slot, src, 64);
          const float vsx = __shfl(S.x, src, 64), vsy = __shfl(S.y, src, 64), vsz = __shfl(S.z, src, 64);
          const float vdx = __shfl(d.x, src, 64), vdy = __shfl(d.y, src, 64), vdz = __shfl(d.z, src, 64);
          const int vwild = __shfl((int)wild, src, 64);
          wave_wild = wave_wild || ballot(thief && vwild != 0) != 0ull; // (wave-uniform)
          if (thief) {
            shared = true;
            slot = vslot;
            S = mk(vsx, vsy, vsz);
            d = mk(vdx, vdy, vdz);
            inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
            wild = vwild != 0;
            best_t = INF;
            best_tri = -1;
            sp = 0;
            sb = 0;
            ref = (uint32_t)got;
            if (pruning) set_delta();
            if (FULLCTR) {
              ray_p0 = ctr.pops;
              ray_t0 = ctr.tris;
              ray_i0 = iters;
              ctr.pops++;
            }
          }
        }
      }
    }

    // ---- inner step for every lane standing on an inner node (P5/fsh:277-302)
    const bool at_inner = (int32_t)ref >= 0;
    if (a.wave_log) {
      const uint32_t ni = (uint32_t)__popcll(ballot(at_inner));
      dbg_inner_lanes += ni;
      dbg_inner_steps += ni ? 1u : 0u;
      dbg_busy_lanes += (uint32_t)__popcll(ballot(ref < REF_DONE));
    }
    if (at_inner) {
      if (FULLCTR) ctr.inner++;
      float4 q0, q1, q2, q3;
      if (ref < (uint32_t)a.lds_nodes) { // top of the tree: staged in LDS
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const v4f lds_v4f;
        lds_v4f* r = (lds_v4f*)(lds_nodes + ref * 5u);
        const v4f w0 = r[0], w1 = r[1], w2 = r[2], w3 = r[3];
        q0 = make_float4(w0.x, w0.y, w0.z, w0.w);
        q1 = make_float4(w1.x, w1.y, w1.z, w1.w);
        q2 = make_float4(w2.x, w2.y, w2.z, w2.w);
        q3 = make_float4(w3.x, w3.y, w3.z, w3.w);
      } else {
        const float4* r = inner + (size_t)ref * 4;
        q0 = r[0];
        q1 = r[1];
        q2 = r[2];
        q3 = r[3];
      }
      float d1, d2;
      if (rel) { // (wave-uniform) boxes relative to the common ray origin: no subtraction here
        if (wave_wild) {
          d1 = hit_aabb_rel(inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
          d2 = hit_aabb_rel(inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
        } else {
          d1 = hit_aabb_tame_rel(inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
          d2 = hit_aabb_tame_rel(inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
        }
      } else if (wave_wild) { // some lane's ray has a zero/NaN direction component: exact select form
        d1 = hit_aabb(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
        d2 = hit_aabb(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
      } else if (pruning) { // (tame rays, absolute boxes: the redo launches)
        float e1, e2;
        d1 = hit_aabb_tame_e(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), e1);
        d2 = hit_aabb_tame_e(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w), e2);
        // a child provably beyond the best hit is treated as missed (flagged children -- slivers below them -- never)
        if (e1 > prune_t && __float_as_uint(q3.z) == 0u) d1 = -1.0f;
        if (e2 > prune_t && __float_as_uint(q3.w) == 0u) d2 = -1.0f;
      } else {
        d1 = hit_aabb_tame(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
        d2 = hit_aabb_tame(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
      }
      const uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
      const bool h1 = d1 > 0.0f, h2 = d2 > 0.0f;
      if (h1 && h2) {
        const bool lf = d1 < d2; // left first; on a tie the right child goes first
        stack[sp * BLOCK] = (int)(lf ? right : left);
        sp++;
        ref = lf ? left : right;
        if (FULLCTR) ctr.pops++;
      } else if (h1 || h2) {
        ref = h1 ? left : right;
        if (FULLCTR) ctr.pops++;
      } else if (sp > sb) {
        sp--;
        ref = (uint32_t)stack[sp * BLOCK];
        if (FULLCTR) ctr.pops++;
      } else {
        finish();
      }
    }

    // ---- leaf phase (hitArray, P5/fsh:238-251): postponed until enough lanes wait at a leaf, or
    // nobody can step.  (Issuing the node and triangle fetches of one iteration together was tried:
    // +22 VGPRs cost a wave per SIMD and 10 % -- see DESIGN.md.)
    const bool at_leaf = (int32_t)ref < -2; // bit 31 set, not REF_NONE / REF_DONE
    const unsigned long long lm = ballot(at_leaf);
    if (lm) {
      const int Lc = (int)__popcll(lm);
      const bool go = Lc >= a.leaf_threshold || !ballot((int32_t)ref >= 0);
      if (go) {
        if (!FULLCTR && Lc <= 32) {
          // cooperative: g = 64 / Lc lanes (power of two) per waiting ray
          const int sh = (Lc <= 1) ? 0 : (32 - __clz(Lc - 1));
          const int g = 64 >> sh;
          const uint32_t rank = lane_rank(lm);
          if (at_leaf) wsrc[rank] = lane;
          __builtin_amdgcn_wave_barrier();
          const int grp = lane >> (6 - sh), m = lane & (g - 1);
          const bool helper = grp < Lc;
          if (a.wave_log) {
            dbg_leaf_lanes += (uint32_t)Lc;
            dbg_leaf_rounds++;
          }
          const int src = helper ? wsrc[grp] : lane;
          const f3 cS = mk(__shfl(S.x, src, 64), __shfl(S.y, src, 64), __shfl(S.z, src, 64));
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
          if (at_leaf && mine != ~0ull) {
            const float t = __uint_as_float((uint32_t)(mine >> 32));
            if (t < best_t) {
              best_t = t;
              best_tri = (int32_t)(uint32_t)mine;
              if (pruning) prune_t = (t + pdelta) * PRUNE_REL_B;
            }
          }
        } else if (at_leaf) {
          const int first = (int)(ref & 0x00ffffffu);
          const int n = (int)((ref >> 24) & 0x7fu) + 1;
          float leaf_best = INF; // hitArray's local res: only the M counter needs it
          for (int i = first; i < first + n; i++) {
            float t;
            const bool hit = hit_triangle_t(sc.tri_geom + (size_t)i * 3, S, d, t);
            if (FULLCTR) {
              ctr.tris++;
              if (hit && t < leaf_best) {
                leaf_best = t;
                ctr.mats++;
              }
            }
            if (hit && t < best_t) {
              best_t = t;
              best_tri = i;
              if (pruning) prune_t = (t + pdelta) * PRUNE_REL_B;
            }
          }
        }
        if (at_leaf) {
          if (sp > sb) {
            sp--;
            ref = (uint32_t)stack[sp * BLOCK];
            if (FULLCTR) ctr.pops++;
          } else {
            finish();
          }
        }
      }
    }
  }

  const unsigned long long rr = wave_sum(ctr.rays);
  if (lane == 0 && rr) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_RAYS], rr);
  if (a.wave_log && lane == 0) {
    unsigned long long* w = a.wave_log + (size_t)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 8;
    w[0] = t_start;
    w[1] = wall_clock64();
    w[2] = wave_iters | ((unsigned long long)dbg_inner_steps << 32);
    w[3] = rr | ((unsigned long long)dbg_inner_lanes << 32);
    w[4] = dbg_leaf_lanes | ((unsigned long long)dbg_leaf_rounds << 32);
    w[5] = dbg_busy_lanes;
  }
  if (FULLCTR) {
    const unsigned long long v1 = wave_sum(ctr.pops), v2 = wave_sum(ctr.inner), v3 = wave_sum(ctr.tris),
                             v4 = wave_sum(ctr.mats);
    if (lane == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_NODE_POPS], v1);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_INNER_POPS], v2);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_TRI_TESTS], v3);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_MAT_FETCH], v4);
    }
  }
}

} // namespace ezd
