// ezrt_tracepk.h -- tracepk_kernel: packet traversal for the PRIMARY rays (stage 0).
//
// A wavefront owns the 64 primary rays of one 8x8 pixel tile of one frame.  They walk the tree
// TOGETHER: one wave-uniform stack of (node, lane mask) entries, node and triangle records fetched
// once per wave at a uniform address (scalar/broadcast loads: no divergent 16-B gathers, the thing
// the per-lane kernel spends its time on), every lane testing its own ray against the shared record.
// The wave visits the union of the nodes its rays touch, in a majority-vote near-first order.
//
// Why this is still bit-exact: hitBVH is unpruned, so the set of triangles a ray tests does not
// depend on the visit order, and its result is min over that set of t -- order only decides which
// triangle wins an EXACT tie in t (strict <, first found: P5/fsh:245, 273).  A lane that ever sees
// a second triangle with t == its current best raises `ambiguous`; such rays (practically none) are
// appended to a redo list and re-traced by traceq_kernel in the reference's own order.  Every other
// ray has a unique minimum and gets exactly the reference's {t, triangle}.
#pragma once
#include "ezrt_traceq.h"

namespace ezd {

struct TracePkArgs {
  const float4* __restrict__ tri_geom;
  const float4* __restrict__ inner;
  uint32_t root_ref;
  RayQueue rq;
  int2* hits;
  uint32_t n_rays;             // multiple of 64 (n_blocks * 256 * n_frames)
  unsigned long long* counters;
  uint32_t* redo_count;        // device counter of ambiguous rays
  uint32_t* redo_slots;        // their ray slots
  int32_t stack_entries;       // rows of the wave-uniform stack (tree depth + 1)
  int32_t budget;              // steps (nodes + triangles) after which a wave hands its rays to traceq_kernel
  float origin[3];             // every primary ray starts here (rq.o is not stored for them)
  uint32_t* dbg;               // diagnostic: [0] wave inner steps [1] wave triangle steps [2] max steps of a wave
};

__global__ __launch_bounds__(BLOCK) void tracepk_kernel(TracePkArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_pk[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // per wave: stack_entries x {ref, mask.lo, mask.hi}
  int* wstack = lds_pk + wave * a.stack_entries * 3;
  // persistent: a wave strides over the packets, so the per-wave bookkeeping atomics happen once per
  // resident wave, not once per packet (a single counter word only takes ~88 atomics/us chip-wide)
  const uint32_t n_packets = a.n_rays / 64u;
  const uint32_t wave_stride = gridDim.x * (BLOCK / 64);
  unsigned long long ndone_total = 0;
  for (uint32_t packet = blockIdx.x * (BLOCK / 64) + (uint32_t)wave; packet < n_packets; packet += wave_stride) {
  const uint32_t slot = packet * 64u + (uint32_t)lane;
  const float4 rd4 = a.rq.d[slot];
  const float4 ro4 = make_float4(a.origin[0], a.origin[1], a.origin[2], 0.0f);
  const bool valid = rd4.w != 0.0f;
  const f3 S = mk(ro4.x, ro4.y, ro4.z), d = mk(rd4.x, rd4.y, rd4.z);
  const f3 inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
  const bool any_wild = ballot(valid && !ray_is_tame(S, inv)) != 0ull;
  float best_t = INF;
  int32_t best_tri = -1;
  bool ambiguous = false;
  uint32_t dbg_inner = 0, dbg_tri = 0;

  unsigned long long cur_mask = ballot(valid);
  uint32_t cur_ref = a.root_ref;
  int sp = 0;
  const unsigned long long lane_bit = 1ull << lane;
  // A packet serialises the UNION of its rays' node visits (up to ~2000 steps for a tile across the
  // Bunny, against ~80 per ray): past `budget` steps the wave gives up and hands all its rays to the
  // per-lane kernel, which spreads exactly that kind of work over lanes.  Sky/floor tiles (95 % of
  // the primaries) finish far below the budget.
  bool bailout = false;
  if (cur_mask) {
    for (;;) {
      if ((int)(dbg_inner + dbg_tri) > a.budget) {
        bailout = true;
        break;
      }
      const bool act = (cur_mask & lane_bit) != 0ull;
      if (cur_ref & LEAF_BIT) {
        const int first = (int)(cur_ref & 0x00ffffffu);
        const int n = (int)((cur_ref >> 24) & 0x7fu) + 1;
        for (int i = first; i < first + n; i++) { // wave-uniform loop, uniform addresses
          dbg_tri++;
          const float4* g = a.tri_geom + (size_t)i * 3;
          if (act) {
            float t;
            if (hit_triangle_t(g, S, d, t)) {
              if (t < best_t) {
                best_t = t;
                best_tri = i;
              } else if (t == best_t && i != best_tri) {
                ambiguous = true; // exact tie: the reference's visit order decides -> redo
              }
            }
          }
        }
      } else {
        dbg_inner++;
        const float4* r = a.inner + (size_t)cur_ref * 4;
        const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
        bool h1 = false, h2 = false, lf = false;
        if (act) {
          float d1, d2;
          if (any_wild) {
            d1 = hit_aabb(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
            d2 = hit_aabb(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
          } else {
            d1 = hit_aabb_tame(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
            d2 = hit_aabb_tame(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
          }
          h1 = d1 > 0.0f;
          h2 = d2 > 0.0f;
          lf = d1 < d2;
        }
        const unsigned long long m1 = ballot(h1), m2 = ballot(h2);
        const uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
        if (m1 && m2) {
          // both children wanted by someone: majority vote on which goes first (speed only)
          const int nl = (int)__popcll(ballot(h1 && h2 && lf)), nr = (int)__popcll(ballot(h1 && h2 && !lf));
          const bool left_first = nl >= nr;
          const uint32_t far_ref = left_first ? right : left;
          const unsigned long long far_mask = left_first ? m2 : m1;
          if (lane == 0) {
            wstack[sp * 3 + 0] = (int)far_ref;
            wstack[sp * 3 + 1] = (int)(uint32_t)far_mask;
            wstack[sp * 3 + 2] = (int)(uint32_t)(far_mask >> 32);
          }
          sp++;
          cur_ref = left_first ? left : right;
          cur_mask = left_first ? m1 : m2;
          continue;
        } else if (m1 | m2) {
          cur_ref = m1 ? left : right;
          cur_mask = m1 ? m1 : m2;
          continue;
        }
      }
      if (sp == 0) break;
      sp--;
      __builtin_amdgcn_wave_barrier();
      cur_ref = (uint32_t)wstack[sp * 3 + 0];
      cur_mask = (unsigned long long)(uint32_t)wstack[sp * 3 + 1] | ((unsigned long long)(uint32_t)wstack[sp * 3 + 2] << 32);
      // the values are wave-uniform; tell the compiler so the loop control stays scalar
      cur_ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur_ref);
      cur_mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cur_mask) |
                 ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cur_mask >> 32)) << 32);
    }
  }
  if (a.dbg && lane == 0) {
    atomicAdd(a.dbg, dbg_inner);
    atomicAdd(a.dbg + 1, dbg_tri);
    atomicMax(a.dbg + 2, dbg_inner + dbg_tri);
  }
  const bool redo = valid && (bailout || ambiguous);
  if (valid && !redo) a.hits[slot] = make_int2(best_tri, __float_as_int(best_t));
  const unsigned long long rm = ballot(redo);
  if (rm) { // one atomic per wave; slots of a wave stay contiguous in the redo list
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.redo_count, (uint32_t)__popcll(rm));
    base = __shfl(base, 0, 64);
    if (redo) a.redo_slots[base + lane_rank(rm)] = slot;
  }
  ndone_total += (unsigned long long)__popcll(ballot(valid && !redo));
  } // packets
  if (lane == 0 && ndone_total) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_RAYS], ndone_total); // redone rays are counted by traceq
}

} // namespace ezd
