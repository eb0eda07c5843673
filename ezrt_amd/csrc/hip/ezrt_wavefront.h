// ezrt_wavefront.h -- the streaming ("wavefront") form of the trace, the timed path.
//
// The v1 megakernel (ezrt_kernels.h: trace_kernel) keeps a whole path in one lane; rocprof showed
// ~18 % of VALU lanes active (dead paths, wildly different traversal lengths) and ~48 % of wave
// time parked on memory at 4 waves/SIMD.  Here a frame chunk is processed as queues:
//
//   raygen_kernel      one thread per pixel-sample: primary ray -> ray queue 0, RNG state
//   traceq_kernel      PERSISTENT hitBVH over a ray queue.  Lanes pull rays one at a time: a lane
//                      whose ray is finished takes the ray it prefetched while traversing and
//                      prefetches the next, so a wave never idles behind its slowest ray.  Inner
//                      nodes are stepped every iteration; leaves are postponed until enough lanes
//                      wait at a leaf (ballot), so the triangle loop runs with a well-filled exec
//                      mask.  Carries {t, tri} only: ~64 VGPRs => 8 waves/SIMD to hide L2 latency.
//                      Traversal stack per lane in LDS.
//   shade_kernel<I>    consumes the hits of bounce b-1, terminates paths into the sample buffer,
//                      starts bounce b (Sobol/CP or rand sampling, Disney BRDF, env lookups) and
//                      COMPACTS survivors into the next queue with one wave-level ballot + one
//                      atomic per wave -- lanes stay converged across bounces.
//
// Per path the arithmetic and its order are exactly the megakernel's (= the oracle's): only the
// schedule changes.  Queue order is non-deterministic, results are not (every path owns its sample
// slot).  The megakernel stays as the path-audit implementation (ezrt_render_paths).
#pragma once
#include "ezrt_kernels.h"

namespace ezd {

constexpr uint32_t FLAG_SHADOW_SHOT = 1u;  // slot 2i holds a live env shadow ray
constexpr uint32_t FLAG_TERMINATE = 2u;    // NdotL <= 0 (P5/fsh:854): finish after the shadow result
constexpr uint32_t FLAG_PDF_DEAD = 4u;     // pdf_brdf <= 0 (P5/fsh:865): ray shot, then break
constexpr uint32_t TRACE_POOL_MAX = 2048;  // ray indices a wave reserves per atomic (large queues)
constexpr uint32_t TRACE_POOL_MIN = 8;     // ... small queues are spread over every resident wave

struct PathState { // SoA of float4, one slot per live path
  float4* s0; // history.xyz, cosine
  float4* s1; // Lo.xyz, pdf
  float4* s2; // f_r.xyz, bits(sample slot)
  float4* s3; // Le0.xyz, bits(seed)
  float4* s4; // shadow contribution.xyz, bits(flags)          (integrator 51 only)
};
struct RayQueue {
  float4* o; // origin.xyz, -
  float4* d; // dir.xyz, valid (1) / skip (0)
};

struct WfArgs {
  DevScene sc;
  EzrtRenderParams p;
  const int2* blocks;
  int32_t n_blocks;
  uint32_t frame_first;
  uint32_t n_slots;        // n_blocks * 256 * n_frames
  float4* samples;         // [n_slots]
  unsigned long long* counters;
  RayQueue rq_in, rq_out;
  PathState st_in, st_out;
  const int2* hits;        // per ray slot of rq_in: (tri, bits(t))
  const uint32_t* n_in;    // paths in the input queue (device)
  uint32_t* n_out;         // paths in the output queue (device, atomically grown)
  int32_t bounce;          // the bounce this stage starts (0 = consumes the primary hits)
};

EZD void slot_to_pixel(const int2* blocks, int n_blocks, uint32_t slot, uint32_t frame_first, int& x, int& y,
                       uint32_t& frame) {
  uint32_t tid = slot & 255u;
  uint32_t b = slot >> 8;
  uint32_t blk = b % (uint32_t)n_blocks, fk = b / (uint32_t)n_blocks;
  int2 org = blocks[blk];
  uint32_t wave = tid >> 6, lane = tid & 63u;
  x = org.x + (int)((wave & 1u) * 8u + (lane & 7u));
  y = org.y + (int)((wave >> 1) * 8u + (lane >> 3));
  frame = frame_first + fk;
}

// ---------------------------------------------------------------------------
// raygen: P5/fsh:315-318, 920-925
__global__ __launch_bounds__(BLOCK) void raygen_kernel(WfArgs a) {
  const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
  if (slot >= a.n_slots) return;
  if (slot == 0) *a.n_out = a.n_slots; // stage 0's path count, read by the first trace launch
  int x, y;
  uint32_t frame;
  slot_to_pixel(a.blocks, a.n_blocks, slot, a.frame_first, x, y, frame);
  const EzrtRenderParams& p = a.p;
  if (!pixel_owned(p, x, y)) {
    a.rq_out.d[slot] = make_float4(0, 0, 0, 0.0f);
    return;
  }
  const uint32_t ix = (uint32_t)x, iy = (uint32_t)y;
  uint32_t seed = (ix * 1973u + iy * 9277u + frame * 26699u) | 1u;
  const float W = (float)p.width, H = (float)p.height;
  float pixx = ((float)ix + 0.5f) / W * 2.0f - 1.0f;
  float pixy = ((float)iy + 0.5f) / H * 2.0f - 1.0f;
  float aax = (rnd(seed) - 0.5f) / W;
  float aay = (rnd(seed) - 0.5f) / H;
  float vx = pixx + aax, vy = pixy + aay, vz = -1.5f;
  const float* m = p.camera_rotate;
  f3 dir = mk(m[0] * vx + m[4] * vy + m[8] * vz, m[1] * vx + m[5] * vy + m[9] * vz, m[2] * vx + m[6] * vy + m[10] * vz);
  dir = normalize(dir);
  a.rq_out.o[slot] = make_float4(p.eye[0], p.eye[1], p.eye[2], 0.0f);
  a.rq_out.d[slot] = make_float4(dir.x, dir.y, dir.z, 1.0f);
  a.st_out.s3[slot] = make_float4(0, 0, 0, __uint_as_float(seed));
}

// ---------------------------------------------------------------------------
// persistent queue traversal
struct TraceQArgs {
  DevScene sc;
  RayQueue rq;
  int2* hits;
  const uint32_t* n_paths; // device count; rays = n_paths * rays_per_path
  uint32_t rays_per_path;
  uint32_t* head;          // queue head (device, zeroed per launch)
  unsigned long long* counters;
  int32_t leaf_threshold;  // lanes waiting at a leaf that trigger the triangle phase
  uint32_t pool_div, pool_max; // pool = clamp(n_rays / (n_waves * pool_div), 8, pool_max)
  int32_t stack_entries;   // LDS stack rows (tree depth); the per-wave lane table follows them
  uint32_t* dbg;           // diagnostic (FULLCTR only): [0] max pops/ray [1] max tris/ray [2] max loop iterations/ray
};

EZD uint32_t lane_rank(unsigned long long mask) { // number of set bits below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool FULLCTR, int WPS>
__global__ __launch_bounds__(BLOCK, WPS) void traceq_kernel(TraceQArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_stack[];
  int* stack = lds_stack + threadIdx.x;
  const DevScene& sc = a.sc;
  const uint32_t n_rays = (*a.n_paths) * a.rays_per_path;
  const int lane = threadIdx.x & 63;
  // reservation size: a short queue (late bounces: few, deep rays) is dealt out in small pools so
  // that every resident wave gets a few lanes of work instead of a few waves getting all of it
  const uint32_t n_waves = gridDim.x * (BLOCK / 64);
  // (one queue-head word sustains only ~88 atomics/us chip-wide -- MI355X_MICROARCH.md "dequeue" --
  // so a wave takes ~1/4 of its fair share per atomic: <= ~4 atomics per wave per launch)
  uint32_t pool_size = n_rays / (n_waves * a.pool_div);
  pool_size = pool_size > a.pool_max ? a.pool_max : (pool_size < TRACE_POOL_MIN ? TRACE_POOL_MIN : pool_size);

  // wave-uniform pool of reserved ray indices
  uint32_t pool_next = 0, pool_end = 0;
  bool exhausted = false; // wave-uniform: the queue has no more rays to hand out

  // prefetched next ray of this lane
  bool nx_valid = false;
  uint32_t nx_slot = 0;
  float4 nx_o = make_float4(0, 0, 0, 0), nx_d = make_float4(0, 0, 0, 0);

  // current ray
  bool work = false;
  bool wild = false; // current ray needs the exact NaN-aware slab test
  uint32_t slot = 0;
  f3 S = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
  float best_t = INF;
  int32_t best_tri = -1;
  int sp = 0;
  uint32_t ref = 0;
  float leaf_best = INF;
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t ray_p0 = 0, ray_t0 = 0, ray_i0 = 0, iters = 0;

  for (;;) {
    if (FULLCTR) iters++;
    // ---- refill: lanes without work adopt their prefetched ray, then prefetch another
    const bool want = !work;
    if (__ballot(want)) {
      if (want && nx_valid) {
        nx_valid = false;
        if (nx_d.w != 0.0f) {
          work = true;
          slot = nx_slot;
          S = mk(nx_o.x, nx_o.y, nx_o.z);
          d = mk(nx_d.x, nx_d.y, nx_d.z);
          inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          wild = !ray_is_tame(S, inv);
          best_t = INF;
          best_tri = -1;
          sp = 0;
          ref = sc.root_ref;
          ctr.rays++;
          if (FULLCTR) { ray_p0 = ctr.pops; ray_t0 = ctr.tris; ray_i0 = iters; ctr.pops++; }
        }
      }
      // lanes with an empty prefetch register reserve the next indices
      const bool need = !nx_valid && !exhausted;
      unsigned long long m = __ballot(need);
      if (m) {
        uint32_t cnt = (uint32_t)__popcll(m);
        if (pool_end - pool_next < cnt) { // wave-uniform: top the pool up with one atomic
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(a.head, pool_size);
          base = __shfl(base, 0, 64);
          // hand out what is left of the old pool first, then the head of the new one; a pool
          // smaller than the request leaves some lanes without a prefetch until the next round
          uint32_t left = pool_end - pool_next;
          uint32_t r = lane_rank(m);
          uint32_t idx = (r < left) ? (pool_next + r) : (base + (r - left));
          uint32_t take = cnt - left;
          if (take > pool_size) take = pool_size;
          const bool served = r < left + take;
          pool_next = base + take;
          pool_end = base + pool_size;
          if (need && served) {
            if (idx < n_rays) {
              nx_slot = idx;
              nx_o = a.rq.o[idx];
              nx_d = a.rq.d[idx];
              nx_valid = true;
            }
          }
          if (base >= n_rays) exhausted = true;
        } else {
          uint32_t idx = pool_next + lane_rank(m);
          pool_next += cnt;
          if (need) {
            if (idx < n_rays) {
              nx_slot = idx;
              nx_o = a.rq.o[idx];
              nx_d = a.rq.d[idx];
              nx_valid = true;
            }
          }
          if (pool_next >= n_rays && pool_end >= n_rays) exhausted = true;
        }
      }
    }
    if (!__ballot(work || nx_valid)) break;

    // ---- inner step for every lane standing on an inner node
    const bool at_inner = work && !(ref & LEAF_BIT);
    if (at_inner) {
      if (FULLCTR) ctr.inner++;
      const float4* r = sc.inner + (size_t)ref * 4;
      float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
      float d1, d2;
      if (__ballot(wild)) { // some lane's ray has a zero/NaN direction component: exact select-based min/max
        d1 = hit_aabb(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
        d2 = hit_aabb(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
      } else {
        d1 = hit_aabb_tame(S, inv, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y));
        d2 = hit_aabb_tame(S, inv, mk(q1.z, q1.w, q2.x), mk(q2.y, q2.z, q2.w));
      }
      uint32_t left = __float_as_uint(q3.x), right = __float_as_uint(q3.y);
      bool h1 = d1 > 0.0f, h2 = d2 > 0.0f;
      if (h1 && h2) {
        bool lf = d1 < d2; // left first
        stack[sp * BLOCK] = (int)(lf ? right : left);
        sp++;
        ref = lf ? left : right;
        if (FULLCTR) ctr.pops++;
      } else if (h1 || h2) {
        ref = h1 ? left : right;
        if (FULLCTR) ctr.pops++;
      } else if (sp > 0) {
        sp--;
        ref = (uint32_t)stack[sp * BLOCK];
        if (FULLCTR) ctr.pops++;
      } else { // traversal finished
        a.hits[slot] = make_int2(best_tri, __float_as_int(best_t));
        work = false;
        if (FULLCTR && a.dbg) { atomicMax(a.dbg, ctr.pops - ray_p0); atomicMax(a.dbg + 1, ctr.tris - ray_t0); atomicMax(a.dbg + 2, iters - ray_i0); }
      }
    }

    // ---- leaf phase: postponed until enough lanes wait at a leaf, or nobody can step
    const bool at_leaf = work && (ref & LEAF_BIT);
    unsigned long long lm = __ballot(at_leaf);
    if (lm) {
      bool go = (int)__popcll(lm) >= a.leaf_threshold || !__ballot(work && !(ref & LEAF_BIT));
      if (go) {
        const int Lc = (int)__popcll(lm);
        if (!FULLCTR && Lc <= 32) {
          // Cooperative leaf phase: the Lc waiting rays share the whole wave.  g = 64 / Lc lanes
          // (power of two) work for each ray, lane m of a group testing triangles m, m+g, ... of
          // that ray's leaf, so a leaf of <= g triangles costs ONE dependent round instead of n.
          // The group minimum of (t bits << 32 | triangle index) is the first triangle (in leaf
          // order) with the smallest t -- exactly hitArray's strict-< scan (P5/fsh:242-249).
          const int sh = (Lc <= 1) ? 0 : (32 - __clz(Lc - 1));
          const int g = 64 >> sh;
          int* wsrc = lds_stack + a.stack_entries * BLOCK + (threadIdx.x >> 6) * 64;
          const uint32_t rank = lane_rank(lm);
          if (at_leaf) wsrc[rank] = lane;
          __builtin_amdgcn_wave_barrier();
          const int grp = lane >> (6 - sh), m = lane & (g - 1);
          const bool helper = grp < Lc;
          const int src = helper ? wsrc[grp] : lane;
          const float sx = __shfl(S.x, src, 64), sy = __shfl(S.y, src, 64), sz = __shfl(S.z, src, 64);
          const float dx = __shfl(d.x, src, 64), dy = __shfl(d.y, src, 64), dz = __shfl(d.z, src, 64);
          const uint32_t lref = (uint32_t)__shfl((int)ref, src, 64);
          unsigned long long key = ~0ull;
          if (helper) {
            const int first = (int)(lref & 0x00ffffffu);
            const int n = (int)((lref >> 24) & 0x7fu) + 1;
            for (int k = m; k < n; k += g) {
              float t;
              bool hit = hit_triangle_t(sc.tri_geom + (size_t)(first + k) * 3, mk(sx, sy, sz), mk(dx, dy, dz), t);
              if (hit) {
                unsigned long long k2 = ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)(first + k);
                key = k2 < key ? k2 : key;
              }
            }
          }
          for (int off = 1; off < g; off <<= 1) {
            unsigned long long other = __shfl_xor(key, off, 64);
            key = other < key ? other : key;
          }
          const unsigned long long mine = __shfl(key, (int)(rank << (6 - sh)), 64);
          if (at_leaf && mine != ~0ull) {
            float t = __uint_as_float((uint32_t)(mine >> 32));
            if (t < best_t) {
              best_t = t;
              best_tri = (int32_t)(uint32_t)mine;
            }
          }
        } else if (at_leaf) {
          int first = (int)(ref & 0x00ffffffu);
          int n = (int)((ref >> 24) & 0x7fu) + 1;
          if (FULLCTR) leaf_best = INF;
          for (int i = first; i < first + n; i++) {
            float t;
            bool hit = hit_triangle_t(sc.tri_geom + (size_t)i * 3, S, d, t);
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
            }
          }
        }
        if (at_leaf) {
          if (sp > 0) {
            sp--;
            ref = (uint32_t)stack[sp * BLOCK];
            if (FULLCTR) ctr.pops++;
          } else {
            a.hits[slot] = make_int2(best_tri, __float_as_int(best_t));
            work = false;
            if (FULLCTR && a.dbg) { atomicMax(a.dbg, ctr.pops - ray_p0); atomicMax(a.dbg + 1, ctr.tris - ray_t0); atomicMax(a.dbg + 2, iters - ray_i0); }
          }
        }
      }
    }
  }

  unsigned long long rr = wave_sum(ctr.rays);
  if (lane == 0 && rr) atomicAdd(&a.counters[EZRT_CTR_RAYS], rr);
  if (FULLCTR) {
    unsigned long long v1 = wave_sum(ctr.pops), v2 = wave_sum(ctr.inner), v3 = wave_sum(ctr.tris), v4 = wave_sum(ctr.mats);
    if (lane == 0) {
      atomicAdd(&a.counters[EZRT_CTR_NODE_POPS], v1);
      atomicAdd(&a.counters[EZRT_CTR_INNER_POPS], v2);
      atomicAdd(&a.counters[EZRT_CTR_TRI_TESTS], v3);
      atomicAdd(&a.counters[EZRT_CTR_MAT_FETCH], v4);
    }
  }
}

// ---------------------------------------------------------------------------
// shading stage
constexpr int SHADE_BLOCK = 1024; // 16 waves share ONE queue-tail atomic per iteration (see block_alloc)

// Compaction slot for every lane with `want`: wave ballots -> per-wave counts in LDS -> one
// atomicAdd per 1024-thread workgroup -> wave offsets.  A single queue-tail word only sustains
// ~88 atomics/us, so per-wave atomics (118 k per stage on C2) would cost more than the shading.
EZD uint32_t block_alloc(uint32_t* counter, bool want, uint32_t* lds /* [SHADE_BLOCK/64 + 1] */) {
  const unsigned long long m = __ballot(want);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NW = SHADE_BLOCK / 64;
  __syncthreads(); // previous iteration's readers are done with lds
  if (lane == 0) lds[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < NW; w++) {
      uint32_t c = lds[w];
      lds[w] = total;
      total += c;
    }
    lds[NW] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  return lds[NW] + lds[wave] + lane_rank(m);
}

EZD uint32_t wave_alloc(uint32_t* counter, bool want) {
  unsigned long long m = __ballot(want);
  if (!m) return 0;
  uint32_t base = 0;
  const int leader = __ffsll((long long)m) - 1;
  if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, leader, 64);
  return base + lane_rank(m);
}

template <int INTEG, bool FULLCTR>
__global__ __launch_bounds__(SHADE_BLOCK) void shade_kernel(WfArgs a) {
  __shared__ uint32_t alloc_lds[SHADE_BLOCK / 64 + 1];
  constexpr bool P5TRI = (INTEG >= 50);
  constexpr bool MIS = (INTEG == EZRT_INTEGRATOR_P5_MIS);
  const DevScene& sc = a.sc;
  const EzrtRenderParams& p = a.p;
  const int b = a.bounce;
  const uint32_t n_in = (b == 0) ? a.n_slots : *a.n_in;
  const uint32_t stride = gridDim.x * SHADE_BLOCK;
  const uint32_t n_round = (n_in + stride - 1) / stride * stride; // keep waves whole for the ballots
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t n_samples = 0;

  for (uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x; i < n_round; i += stride) {
    bool live = i < n_in;       // this lane holds a path
    bool emit = false;          // ... that continues into the next queue
    uint32_t sslot = 0, seed = 0, flags = 0;
    f3 history = mk(1, 1, 1), Lo = mk(0, 0, 0), Le0 = mk(0, 0, 0), f_r = mk(0, 0, 0), shadowC = mk(0, 0, 0);
    float cosine = 0.0f, pdf = 1.0f;
    f3 rayL = mk(0, 0, 0), shadowL = mk(0, 0, 0);
    Hit hit;
    hit.P = mk(0, 0, 0);

    // first-level loads: addresses depend on i only, so issue them all up front (one memory
    // round trip) instead of discovering them one branch at a time
    const uint32_t ii = live ? i : 0u;
    const uint32_t rslot = (b == 0 || !MIS) ? ii : (2u * ii + 1u);
    const float4 rd4 = a.rq_in.d[rslot];
    const float4 ro4 = a.rq_in.o[rslot];
    const int2 h = a.hits[rslot];
    const float4 s3 = a.st_in.s3[ii];
    float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s4 = s0;
    int2 sh = make_int2(-1, 0);
    if (b > 0) {
      s0 = a.st_in.s0[ii];
      s1 = a.st_in.s1[ii];
      s2 = a.st_in.s2[ii];
      if (MIS) {
        s4 = a.st_in.s4[ii];
        sh = a.hits[2u * ii];
      }
    }
    bool done = false;
    if (live) {
      f3 colour = mk(0, 0, 0);
      const f3 rd = mk(rd4.x, rd4.y, rd4.z);
      if (b == 0) {
        sslot = i;
        if (rd4.w == 0.0f) {
          live = false; // pixel not owned by this shard
        } else {
          n_samples = n_samples + 1;
          if (h.x < 0) { // primary miss: P5/fsh:931-933
            colour = hdr_color<FULLCTR>(sc, rd, p.env_clamp, ctr);
            done = true;
          } else {
            shade_point<P5TRI>(sc, h.x, __int_as_float(h.y), mk(ro4.x, ro4.y, ro4.z), rd, hit);
            Le0 = hit.m.emissive;
            seed = __float_as_uint(s3.w);
          }
        }
      } else {
        history = mk(s0.x, s0.y, s0.z);
        cosine = s0.w;
        Lo = mk(s1.x, s1.y, s1.z);
        pdf = s1.w;
        f_r = mk(s2.x, s2.y, s2.z);
        sslot = __float_as_uint(s2.w);
        Le0 = mk(s3.x, s3.y, s3.z);
        seed = __float_as_uint(s3.w);
        if (MIS) {
          flags = __float_as_uint(s4.w);
          if (flags & FLAG_SHADOW_SHOT) { // P5/fsh:826-841
            if (sh.x < 0) {
              Lo = Lo + mk(s4.x, s4.y, s4.z);
              if (FULLCTR) {
                ctr.envmap++;
                ctr.envcache++;
              }
            }
          }
        }
        if (MIS && (flags & FLAG_TERMINATE)) {
          done = true;
        } else if (MIS && (flags & FLAG_PDF_DEAD)) {
          done = true;
        } else {
          if (h.x < 0) {
            f3 sky = hdr_color<FULLCTR>(sc, rd, p.env_clamp, ctr);
            if (MIS) {
              float pdf_light = hdr_pdf<FULLCTR>(sc, rd, ctr);
              float w = mis_mix_weight(pdf, pdf_light);
              Lo = Lo + (((history * w) * sky) * f_r) * cosine / pdf;
            } else {
              Lo = Lo + ((history * sky) * f_r) * cosine / pdf;
            }
            done = true;
          } else {
            shade_point<P5TRI>(sc, h.x, __int_as_float(h.y), mk(ro4.x, ro4.y, ro4.z), rd, hit);
            Lo = Lo + ((history * hit.m.emissive) * f_r) * cosine / pdf;
            history = history * (f_r * cosine / pdf);
          }
        }
        if (done) colour = Le0 + Lo;
      }
      if (live && !done && b >= p.max_bounce) {
        colour = Le0 + Lo;
        done = true;
      }
      if (live && done) a.samples[sslot] = make_float4(colour.x, colour.y, colour.z, 1.0f);

      if (live && !done) {
        // ---- start bounce b (loop body of pathTracing*, P5/fsh:767-804 / 815-887)
        int x, y;
        uint32_t frame;
        slot_to_pixel(a.blocks, a.n_blocks, sslot, a.frame_first, x, y, frame);
        const f3 V = -hit.viewDir, N = hit.N;
        flags = 0;
        bool shoot = true;
        if (MIS) {
          float h1 = rnd(seed);
          float h2 = rnd(seed);
          f3 Lh = sample_hdr<FULLCTR>(sc, h1, h2, ctr);
          if (dot(N, Lh) > 0.0f) {
            flags |= FLAG_SHADOW_SHOT;
            shadowL = Lh;
            // contribution if unoccluded, evaluated eagerly (pure functions of the path state)
            Counters dummy = {0, 0, 0, 0, 0, 0, 0};
            f3 color = hdr_color<false>(sc, Lh, p.env_clamp, dummy);
            float pdf_light = hdr_pdf<false>(sc, Lh, dummy);
            f3 fr = brdf_evaluate<false>(V, N, Lh, mk(0, 0, 0), mk(0, 0, 0), hit.m);
            float pdf_brdf = brdf_pdf(V, N, Lh, hit.m);
            float w = mis_mix_weight(pdf_light, pdf_brdf);
            shadowC = (((history * w) * color) * fr) * dot(N, Lh) / pdf_light;
          }
        }
        float xi1, xi2;
        if (INTEG >= 50) {
          float cpu, cpv;
          cp_offsets((uint32_t)x, (uint32_t)y, cpu, cpv);
          const uint32_t gray = gray_code(frame + 1u);
          uint32_t d0 = ((uint32_t)b * 2u) & 7u, d1 = ((uint32_t)b * 2u + 1u) & 7u;
          xi1 = cp_rotate(sobol(d0, gray), cpu);
          xi2 = cp_rotate(sobol(d1, gray), cpv);
        } else {
          xi1 = rnd(seed);
          xi2 = rnd(seed);
        }
        if (MIS) {
          float xi3 = rnd(seed);
          rayL = sample_brdf(xi1, xi2, xi3, V, N, hit.m);
          cosine = dot(N, rayL);
          if (cosine <= 0.0f) {
            shoot = false;
            flags |= FLAG_TERMINATE;
          } else {
            f_r = brdf_evaluate<false>(V, N, rayL, mk(0, 0, 0), mk(0, 0, 0), hit.m);
            pdf = brdf_pdf(V, N, rayL, hit.m);
            if (pdf <= 0.0f) flags |= FLAG_PDF_DEAD;
          }
        } else {
          rayL = to_normal_hemisphere(sample_hemisphere(xi1, xi2), N);
          pdf = 1.0f / (2.0f * PI);
          cosine = ez_max(0.0f, dot(rayL, N));
          if (INTEG == EZRT_INTEGRATOR_P3_DIFFUSE) {
            f_r = hit.m.baseColor / PI;
          } else {
            f3 tangent, bitangent;
            get_tangent(N, tangent, bitangent);
            f_r = brdf_evaluate<INTEG == EZRT_INTEGRATOR_P4_DISNEY>(V, N, rayL, tangent, bitangent, hit.m);
          }
        }
        if (!shoot && !(flags & FLAG_SHADOW_SHOT)) { // nothing pending: the path ends here
          f3 c2 = Le0 + Lo;
          a.samples[sslot] = make_float4(c2.x, c2.y, c2.z, 1.0f);
        } else {
          emit = true;
          if (!shoot) rayL = mk(0, 0, 0);
        }
      }
    }

    // ---- compaction: one ballot + one atomic per wave
    const uint32_t o = block_alloc(a.n_out, emit, alloc_lds);
    if (emit) {
      a.st_out.s0[o] = make_float4(history.x, history.y, history.z, cosine);
      a.st_out.s1[o] = make_float4(Lo.x, Lo.y, Lo.z, pdf);
      a.st_out.s2[o] = make_float4(f_r.x, f_r.y, f_r.z, __uint_as_float(sslot));
      a.st_out.s3[o] = make_float4(Le0.x, Le0.y, Le0.z, __uint_as_float(seed));
      const bool shoot = !(flags & FLAG_TERMINATE);
      if (MIS) {
        a.st_out.s4[o] = make_float4(shadowC.x, shadowC.y, shadowC.z, __uint_as_float(flags));
        a.rq_out.o[2u * o] = make_float4(hit.P.x, hit.P.y, hit.P.z, 0.0f);
        a.rq_out.d[2u * o] = make_float4(shadowL.x, shadowL.y, shadowL.z, (flags & FLAG_SHADOW_SHOT) ? 1.0f : 0.0f);
        a.rq_out.o[2u * o + 1u] = make_float4(hit.P.x, hit.P.y, hit.P.z, 0.0f);
        a.rq_out.d[2u * o + 1u] = make_float4(rayL.x, rayL.y, rayL.z, shoot ? 1.0f : 0.0f);
      } else {
        a.rq_out.o[o] = make_float4(hit.P.x, hit.P.y, hit.P.z, 0.0f);
        a.rq_out.d[o] = make_float4(rayL.x, rayL.y, rayL.z, 1.0f);
      }
    }
  }

  const int lane = threadIdx.x & 63;
  unsigned long long s = wave_sum(n_samples);
  if (lane == 0 && s) atomicAdd(&a.counters[EZRT_CTR_SAMPLES], s);
  if (FULLCTR) {
    unsigned long long v5 = wave_sum(ctr.envmap), v6 = wave_sum(ctr.envcache);
    if (lane == 0) {
      atomicAdd(&a.counters[EZRT_CTR_ENV_MAP], v5);
      atomicAdd(&a.counters[EZRT_CTR_ENV_CACHE], v6);
    }
  }
}

} // namespace ezd
