// ezrt_wavefront.h -- the streaming ("wavefront") form of the trace, the timed path.
//
// The v1 megakernel (ezrt_kernels.h: trace_kernel) keeps a whole path in one lane; rocprof showed
// ~18 % of VALU lanes active (dead paths, wildly different traversal lengths) and ~48 % of wave
// time parked on memory at 4 waves/SIMD.  Here a frame chunk is processed as queues:
//
//   raygen_kernel      one thread per pixel-sample: primary ray direction -> ray queue 0, in a
//                      scattered 8x8 sub-block order (queue_to_sample)
//   traceq_kernel      PERSISTENT hitBVH over a ray queue (ezrt_traceq.h): per-lane ray refill,
//                      LDS traversal stack + LDS-resident top of the tree, postponed cooperative
//                      leaves, intra-wave work stealing.  Carries {t, tri} only.
//   shade_kernel<I>    consumes the hits of bounce b-1, terminates paths into the sample buffer,
//                      starts bounce b (Sobol/CP or rand sampling, Disney BRDF, env lookups) and
//                      COMPACTS survivors into the next queue with one ballot per wave + one atomic
//                      per workgroup -- lanes stay converged across bounces.
//
// Per path the arithmetic and its order are exactly the megakernel's (= the oracle's): only the
// schedule changes.  Queue order is non-deterministic, results are not (every path owns its sample
// slot).  The megakernel stays as the path-audit implementation (ezrt_render_paths).
#pragma once
#include "ezrt_kernels.h"
#include "ezrt_traceq.h"
#include "ezrt_traceq4.h"

namespace ezd {

constexpr uint32_t FLAG_SHADOW_SHOT = 1u;  // slot 2i holds a live env shadow ray
constexpr uint32_t FLAG_TERMINATE = 2u;    // NdotL <= 0 (P5/fsh:854): finish after the shadow result
constexpr uint32_t FLAG_PDF_DEAD = 4u;     // pdf_brdf <= 0 (P5/fsh:865): ray shot, then break
struct PathState { // SoA of float4, one slot per live path
  float4* s0; // history.xyz, cosine
  float4* s1; // Lo.xyz, pdf
  float4* s2; // f_r.xyz, bits(sample slot)
  float4* s3; // Le0.xyz, bits(seed)
  float4* s4; // shadow contribution.xyz, bits(flags)          (integrator 51 only)
  // Integrator 50 (pdf is the constant 1/(2 pi), no RNG state after ray generation, Le0 = the emission of the
  // primary hit's triangle) carries a COMPACT state instead -- the shading stages are bound by streaming this state:
  //   into stage 1 (history = 1 and Lo = 0 exactly):  s0 = f_r.xyz, cosine   s1 = float2: bits(sample slot), bits(tri0)   24 B
  //   (tri0 = the primary hit's triangle if its emission is not all-zero bits, else 0xffffffff)
  //   into stages >= 2:  s0 = history.xyz, cosine   s1 = Lo.xyz, bits(sample slot)   s2 = f_r.xyz, bits(tri0)      48 B
  // (tri0 = the primary hit's triangle; Le0 is re-read from its material at the path's end)
  // The MIS integrators (51, 52) carry the general state from stage 1 on, but INTO stage 1 -- the biggest stage, and its
  // first pass is bound by streaming this state -- history = 1 and Lo = 0 are known as well:
  //   s0 = f_r.xyz, cosine   s1 = pdf, bits(sample slot), bits(seed), bits(tri0)   s4 as above          48 B instead of 80
};
template <int INTEG>
constexpr bool compact_state() { return INTEG == EZRT_INTEGRATOR_P5_SOBOL; }
template <int INTEG, int STAGE>
constexpr bool mis_stage1_state() { return integ_mis<INTEG>() && STAGE == 1; } // (see PathState)
struct WfArgs {
  DevScene sc;
  EzrtRenderParams p;
  const int2* blocks;
  int32_t n_blocks;
  uint32_t frame_first;
  uint32_t n_slots;        // n_blocks * 256 * n_frames
  Sample3* samples;        // [n_slots]
  unsigned long long* counters;
  RayQueue rq_in, rq_out;
  PathState st_in, st_out;
  const int2* hits;        // per ray slot of rq_in: (tri, bits(t))
  unsigned long long* hits_out; // hit records of rq_out's slots (ping-pong with `hits`; written by the trace)
  const uint32_t* n_in;    // paths in the input queue (device)
  uint32_t* n_out;         // paths in the output queue (device, atomically grown)
  int32_t bounce;          // the bounce this stage starts (0 = consumes the primary hits)
  const float* sobol_tab;  // [frame - frame_first][16]: sobol(d, grayCode(frame + 1)), filled by raygen_kernel
  float* sobol_out;        // (the same table, as raygen_kernel writes it)
  uint32_t n_frames;       // frames of the chunk
  uint4* defer_list;       // split shading: per workgroup of shade_miss_kernel, the paths with a surface interaction: (queue
                           // position, hit triangle, bits(t), shadow ray's hit triangle) -- the second pass starts from the entry,
                           // not from another dependent read of the hit records (HIT_PENDING entries excepted: those it re-reads)
  uint32_t* defer_count;   // ... and how many; shaded by the same workgroup index of shade_hit_kernel
  FastDiv div_blocks;      // division by n_blocks
  FastDiv div_sub;         // division by the queue granules per frame, (n_blocks * 256) >> scatter_shift
  uint32_t scatter;        // queue order of the primary rays: see queue_to_sample (1 = identity)
  uint32_t scatter_shift;  // scattered granule = 1 << scatter_shift slots (6: 8x8 sub-block, 8: 16x16 block, 5: 8x4 pixels)
  uint32_t all_owned;      // 1: every queue position of stage 0 is a pixel-sample of this call (one shard, whole 16x16 blocks)
  uint32_t gen_primary;    // 1: queue 0 holds no directions -- stage 0 recomputes its ray from the queue position (primary_dir)
};

// ---------------------------------------------------------------------------
// raygen: P5/fsh:315-318, 920-925
__global__ __launch_bounds__(BLOCK) void raygen_kernel(WfArgs a, ChunkPrologue g) {
  const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;
  chunk_prologue(g, slot, gridDim.x * BLOCK);
  if (slot >= a.n_slots) return;
  if (slot == 0) *a.n_out = a.n_slots; // stage 0's path count, read by the first trace launch
  if (blockIdx.x == 0) // the chunk's Sobol table (read by the shading stages): sobol(d, grayCode(frame + 1)), 16 dims per frame
    for (uint32_t k = threadIdx.x; k < a.n_frames * 16u; k += BLOCK)
      a.sobol_out[k] = sobol(k & 15u, gray_code(a.frame_first + (k >> 4) + 1u));
  // only the direction is stored: every primary ray starts at the eye (the trace takes it from its
  // arguments) and the shading stage re-derives the RNG state from the slot (two hashes) -- 32 B per
  // pixel-sample less to write here and 48 B less to read downstream
  a.rq_out.d[slot] = primary_dir(a.p, a.blocks, a.div_blocks, a.div_sub, a.scatter, a.scatter_shift, a.frame_first, slot);
}

// The chunk's housekeeping alone, for chunks whose primary rays are generated where they are consumed (gen_primary):
// Sobol table, zeroed queue heads and stage counters, the eye-relative records, stage 0's path count.
__global__ __launch_bounds__(BLOCK) void chunk_prologue_kernel(WfArgs a, ChunkPrologue g) {
  const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
  chunk_prologue(g, tid, gridDim.x * BLOCK);
  if (tid == 0) *a.n_out = a.n_slots;
  if (blockIdx.x == 0)
    for (uint32_t k = threadIdx.x; k < a.n_frames * 16u; k += BLOCK)
      a.sobol_out[k] = sobol(k & 15u, gray_code(a.frame_first + (k >> 4) + 1u));
}

// ---------------------------------------------------------------------------
// shading stage
constexpr int SHADE_MISS_WAVES = 8; // waves/SIMD the miss half of the split shading is compiled for (<= 64 VGPRs)
constexpr int SHADE_BLOCK = 512; // 8 waves share ONE queue-tail atomic per iteration (see block_alloc); two workgroups
                                 // per CU, so one computes while the other sits in its barriers (1024: -3 %, 256: atomic-bound)

// Compaction slot for every lane with `want`: wave ballots -> per-wave counts in LDS -> one
// atomicAdd per 1024-thread workgroup -> wave offsets.  A single queue-tail word only sustains
// ~88 atomics/us, so per-wave atomics (118 k per stage on C2) would cost more than the shading.
EZD uint32_t block_alloc(uint32_t* counter, bool want, uint32_t* lds /* [SHADE_BLOCK/64 + 1] */) {
  const unsigned long long m = ballot(want);
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
  unsigned long long m = ballot(want);
  if (!m) return 0;
  uint32_t base = 0;
  const int leader = __ffsll((long long)m) - 1;
  if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, leader, 64);
  return base + lane_rank(m);
}

// Block-wide exclusive rank of the lanes with `want` (no global atomic); total = their number.
EZD uint32_t block_rank(bool want, uint32_t* lds /* [SHADE_BLOCK/64 + 1] */, uint32_t& total) {
  const unsigned long long m = ballot(want);
  const int wave = threadIdx.x >> 6;
  constexpr int NW = SHADE_BLOCK / 64;
  __syncthreads(); // previous readers are done with lds
  if ((threadIdx.x & 63) == 0) lds[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < NW; w++) {
      uint32_t c = lds[w];
      lds[w] = t;
      t += c;
    }
    lds[NW] = t;
  }
  __syncthreads();
  total = lds[NW];
  return lds[wave] + lane_rank(m);
}

// What one path hands to the compaction at the end of a shading stage.
struct ShadeOut {
  bool emit;
  uint32_t sslot, seed, flags, tri0;
  f3 history, Lo, Le0, f_r, shadowC, rayL, shadowL, P;
  float cosine, pdf;
};

// One path through stage b (STAGE: 0 = b == 0, the path has no state yet; 1 = b == 1; 2 = b >= 2).  PASS 0: everything inline.
// PASS 1: everything but the surface interaction -- a path whose ray hit a triangle and that
// is still alive returns true ("deferred") and touches nothing.  PASS 2: the deferred paths, regrouped
// densely by the caller.  Bounce rays mostly leave the scene (90 % on C2), so without the regrouping
// every wave ran the ~700-instruction surface code for a handful of its lanes.
struct ShadeIn { // what stage b reads for one path (from the queues)
  float4 rd4, ro4, s0, s1, s2, s3, s4;
  int2 h, sh;
};
// entry (PASS 2 of the split kernels, or NULL): the list entry shade_miss_kernel wrote for this path
template <int INTEG, int PASS, int STAGE>
EZD void shade_load(const WfArgs& a, const uint32_t i, const bool live, ShadeIn& in, const uint4* entry = nullptr) {
  constexpr bool MIS = integ_mis<INTEG>();
  constexpr bool B0 = (STAGE == 0);
  constexpr bool COMPACT = compact_state<INTEG>();
  constexpr bool MIS1 = mis_stage1_state<INTEG, STAGE>();
  const EzrtRenderParams& p = a.p;
  const int b = a.bounce;
  // first-level loads: addresses depend on i only, so issue them all up front (one memory
  // round trip) instead of discovering them one branch at a time
  const uint32_t ii = live ? i : 0u;
  const uint32_t rslot = (b == 0 || !MIS) ? ii : (2u * ii + 1u);
  const bool from_entry = PASS == 2 && entry && (int32_t)entry->y != HIT_PENDING && (!MIS || (int32_t)entry->w != HIT_PENDING);
  int2 h = from_entry ? make_int2((int32_t)entry->y, (int32_t)entry->z) : a.hits[rslot];
  float4 rd4;
  if (B0 && a.gen_primary) {
    rd4 = primary_dir(a.p, a.blocks, a.div_blocks, a.div_sub, a.scatter, a.scatter_shift, a.frame_first, rslot);
  } else if (B0 && PASS == 1 && a.all_owned) {
    // the first pass of the primary stage only looks at the direction of a ray that left the scene (the env lookup): 40 %
    // of C2's 268 MB of directions are not read here (a second round trip for the others; the kernel is HBM-bound)
    asm volatile("" : "+v"(h.x), "+v"(h.y));
    rd4 = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (h.x < 0) rd4 = a.rq_in.d[rslot];
  } else {
    rd4 = a.rq_in.d[rslot];
  }
  float4 ro4 = make_float4(p.eye[0], p.eye[1], p.eye[2], 0.0f);
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0, s4 = s0;
  int2 sh = make_int2(-1, 0);
  if (!B0) {
    if (PASS != 1) ro4 = a.rq_in.o[ii]; // (only a surface interaction needs the ray origin; one per path)
    if (!COMPACT && !MIS1) s3 = a.st_in.s3[ii];
    s0 = a.st_in.s0[ii];
    if (COMPACT && STAGE == 1) { // (bits(sample slot), bits(tri0)) only: 8-byte records in the s1 array
      const float2 q = reinterpret_cast<const float2*>(a.st_in.s1)[ii];
      s1 = make_float4(q.x, q.y, 0.0f, 0.0f);
    } else {
      s1 = a.st_in.s1[ii];
    }
    if ((!COMPACT || STAGE >= 2) && !MIS1) s2 = a.st_in.s2[ii];
    if (MIS) {
      s4 = a.st_in.s4[ii];
      sh = from_entry ? make_int2((int32_t)entry->w, 0) : a.hits[2u * ii];
    }
  }
  // Pin the loads here: left alone, the compiler sinks every one of them into the branch that first uses
  // it and narrows it to the component used there (rd4.w, then h.x, then rd4.xyz ...), i.e. a chain of
  // three or four dependent memory round trips per path instead of one.
#define EZ_PIN4(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
  EZ_PIN4(rd4);
  asm volatile("" : "+v"(h.x), "+v"(h.y));
  if (!B0) {
    if (PASS != 1) EZ_PIN4(ro4);
    EZ_PIN4(s0);
    EZ_PIN4(s1);
    if ((!COMPACT || STAGE >= 2) && !MIS1) EZ_PIN4(s2);
    if (!COMPACT && !MIS1) EZ_PIN4(s3);
    if (MIS) {
      EZ_PIN4(s4);
      asm volatile("" : "+v"(sh.x), "+v"(sh.y));
    }
  }
#undef EZ_PIN4
  in.rd4 = rd4;
  in.ro4 = ro4;
  in.s0 = s0;
  in.s1 = s1;
  in.s2 = s2;
  in.s3 = s3;
  in.s4 = s4;
  in.h = h;
  in.sh = sh;
}

// `i` is only used by stage 0 (queue position -> sample slot)
template <int INTEG, bool FULLCTR, int PASS, int STAGE>
EZD bool shade_body(const WfArgs& a, const uint32_t i, bool live, const ShadeIn& in, Counters& ctr, uint32_t& n_samples,
                    ShadeOut& o) {
  constexpr bool P5TRI = (INTEG >= 50);
  constexpr bool MIS = integ_mis<INTEG>();
  const DevScene& sc = a.sc;
  const EzrtRenderParams& p = a.p;
  const int b = a.bounce;
  constexpr bool B0 = (STAGE == 0);
  constexpr bool COMPACT = compact_state<INTEG>();
  uint32_t sslot = 0, seed = 0, flags = 0, tri0 = 0;
  f3 history = mk(1, 1, 1), Lo = mk(0, 0, 0), Le0 = mk(0, 0, 0), f_r = mk(0, 0, 0), shadowC = mk(0, 0, 0);
  float cosine = 0.0f, pdf = 1.0f;
  f3 rayL = mk(0, 0, 0), shadowL = mk(0, 0, 0);
  Hit hit;
  hit.P = mk(0, 0, 0);
  o.emit = false;
  const float4 rd4 = in.rd4, ro4 = in.ro4, s0 = in.s0, s1 = in.s1, s2 = in.s2, s3 = in.s3, s4 = in.s4;
  const int2 h = in.h, sh = in.sh;
  bool done = false;
  if (!live) return false;
  f3 colour = mk(0, 0, 0);
  const f3 rd = mk(rd4.x, rd4.y, rd4.z);
  if (B0) {
    sslot = queue_to_sample(i, a.div_sub, a.scatter, a.scatter_shift);
    if (rd4.w == 0.0f) {
      live = false; // pixel not owned by this shard
    } else {
      if (PASS != 2) n_samples = n_samples + 1;
      if (PASS == 1 && h.x == HIT_PENDING) return true; // answer still with the redo launch: decided in the second pass
      if (h.x < 0) { // primary miss: P5/fsh:931-933
        colour = hdr_color<FULLCTR>(sc, rd, p.env_clamp, ctr);
        done = true;
      } else {
        if (PASS == 1) return true; // surface interaction: shaded in dense waves (shade_hit_kernel)
        shade_point<P5TRI>(sc, h.x, __int_as_float(h.y), mk(ro4.x, ro4.y, ro4.z), rd, hit);
        Le0 = hit.m.emissive;
        // compact state: the primary hit's triangle is carried only when its emission has a set bit -- a later stage
        // then re-reads Le0 from its material, and every other path (all but the light's pixels) skips that gather
        tri0 = (__float_as_uint(Le0.x) | __float_as_uint(Le0.y) | __float_as_uint(Le0.z)) ? (uint32_t)h.x : 0xffffffffu;
        if (INTEG != EZRT_INTEGRATOR_P5_SOBOL) { // (integrator 50 draws nothing after ray generation: Sobol + CP only)
          // RNG state after the two anti-aliasing draws of ray generation (P5/fsh:315-318, 920-921)
          int x0, y0;
          uint32_t f0;
          slot_to_pixel(a.blocks, a.div_blocks, sslot, a.frame_first, x0, y0, f0);
          seed = ((uint32_t)x0 * 1973u + (uint32_t)y0 * 9277u + f0 * 26699u) | 1u;
          wang_hash(seed);
          wang_hash(seed);
        }
      }
    }
  } else {
    // a slot the previous stage's second pass left empty (a path that was pending there and then left the scene): it
    // holds no ray (w = 0), no hit record and no state -- nothing below may be read
    if (!MIS && rd4.w == 0.0f) return false;
    if (COMPACT) {
      cosine = s0.w;
      pdf = 1.0f / (2.0f * PI); // (what the bounce wrote: P5/fsh:774)
      if (STAGE == 1) { // history = (1,1,1), Lo = (0,0,0): the initial values, untouched by bounce 0
        f_r = mk(s0.x, s0.y, s0.z);
        sslot = __float_as_uint(s1.x);
        tri0 = __float_as_uint(s1.y);
      } else {
        history = mk(s0.x, s0.y, s0.z);
        Lo = mk(s1.x, s1.y, s1.z);
        sslot = __float_as_uint(s1.w);
        f_r = mk(s2.x, s2.y, s2.z);
        tri0 = __float_as_uint(s2.w);
      }
      // Le0 = the primary hit's emission (getMaterial: texel 6 of its record), needed when the path ends; all-zero
      // bits (tri0 = none) for every non-emissive primary hit
      if (tri0 != 0xffffffffu) {
        const float* e = sc.tri_ref + (size_t)tri0 * EZRT_TRI_FLOATS + 18;
        Le0 = mk(e[0], e[1], e[2]);
      }
    } else if (mis_stage1_state<INTEG, STAGE>()) { // history = (1,1,1), Lo = (0,0,0): the initial values, untouched by bounce 0
      f_r = mk(s0.x, s0.y, s0.z);
      cosine = s0.w;
      pdf = s1.x;
      sslot = __float_as_uint(s1.y);
      seed = __float_as_uint(s1.z);
      tri0 = __float_as_uint(s1.w);
      if (tri0 != 0xffffffffu) { // (as for the compact state above)
        const float* e = sc.tri_ref + (size_t)tri0 * EZRT_TRI_FLOATS + 18;
        Le0 = mk(e[0], e[1], e[2]);
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
    }
    if (PASS == 1 && (h.x == HIT_PENDING || (MIS && sh.x == HIT_PENDING))) return true; // (see HIT_PENDING)
    if (MIS) {
      flags = __float_as_uint(s4.w);
      if (flags & FLAG_SHADOW_SHOT) { // P5/fsh:826-841
        if (sh.x < 0) {
          Lo = Lo + mk(s4.x, s4.y, s4.z);
          if (FULLCTR && PASS != 2) {
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
        f3 sky;
        float pdf_light = 0.0f;
        if (MIS) hdr_color_pdf<FULLCTR>(sc, rd, p.env_clamp, ctr, sky, pdf_light);
        else sky = hdr_color<FULLCTR>(sc, rd, p.env_clamp, ctr);
        if (MIS) {
          float w = mis_mix_weight(pdf, pdf_light);
          Lo = Lo + (((history * w) * sky) * f_r) * cosine / pdf;
        } else {
          Lo = Lo + ((history * sky) * f_r) * cosine / pdf;
        }
        done = true;
      } else {
        if (PASS == 1) return true; // surface interaction: regrouped into dense waves by the caller
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
  if (live && done) a.samples[sslot] = Sample3{colour.x, colour.y, colour.z};

  if (live && !done) {
    // ---- start bounce b (loop body of pathTracing*, P5/fsh:767-804 / 815-887)
    int x, y;
    uint32_t frame;
    slot_to_pixel(a.blocks, a.div_blocks, sslot, a.frame_first, x, y, frame);
    const f3 V = -hit.viewDir, N = hit.N;
    constexpr bool ANISO_IS = integ_aniso_is<INTEG>();
    f3 X = mk(0, 0, 0), Y = mk(0, 0, 0);
    if (ANISO_IS) get_tangent(N, X, Y);
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
        f3 color;
        float pdf_light;
        hdr_color_pdf<false>(sc, Lh, p.env_clamp, dummy, color, pdf_light);
        f3 fr;
        float pdf_brdf;
        brdf_evaluate_pdf<ANISO_IS>(V, N, Lh, X, Y, hit.m, fr, pdf_brdf);
        float w = mis_mix_weight(pdf_light, pdf_brdf);
        shadowC = (((history * w) * color) * fr) * dot(N, Lh) / pdf_light;
      }
    }
    float xi1, xi2;
    if (INTEG >= 50) {
      float cpu, cpv;
      cp_offsets((uint32_t)x, (uint32_t)y, cpu, cpv);
      const float* sob = a.sobol_tab + (size_t)(frame - a.frame_first) * 16u; // sobol(d, grayCode(frame + 1))
      const uint32_t d0 = ((uint32_t)b * 2u) & sc.sobol_mask, d1 = ((uint32_t)b * 2u + 1u) & sc.sobol_mask;
      xi1 = cp_rotate(sob[d0], cpu);
      xi2 = cp_rotate(sob[d1], cpv);
    } else {
      xi1 = rnd(seed);
      xi2 = rnd(seed);
    }
    if (MIS) {
      float xi3 = rnd(seed);
      rayL = ANISO_IS ? sample_brdf_aniso(xi1, xi2, xi3, V, N, X, Y, hit.m) : sample_brdf(xi1, xi2, xi3, V, N, hit.m);
      cosine = dot(N, rayL);
      if (cosine <= 0.0f) {
        shoot = false;
        flags |= FLAG_TERMINATE;
      } else {
        brdf_evaluate_pdf<ANISO_IS>(V, N, rayL, X, Y, hit.m, f_r, pdf);
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
      a.samples[sslot] = Sample3{c2.x, c2.y, c2.z};
    } else {
      o.emit = true;
      if (!shoot) rayL = mk(0, 0, 0);
    }
  }
  o.sslot = sslot;
  o.tri0 = tri0;
  o.seed = seed;
  o.flags = flags;
  o.history = history;
  o.Lo = Lo;
  o.Le0 = Le0;
  o.f_r = f_r;
  o.shadowC = shadowC;
  o.rayL = rayL;
  o.shadowL = shadowL;
  o.P = hit.P;
  o.cosine = cosine;
  o.pdf = pdf;
  return false;
}
template <int INTEG, bool FULLCTR, int PASS, int STAGE>
EZD bool shade_path(const WfArgs& a, const uint32_t i, bool live, Counters& ctr, uint32_t& n_samples, ShadeOut& o) {
  ShadeIn in;
  shade_load<INTEG, PASS, STAGE>(a, i, live, in);
  return shade_body<INTEG, FULLCTR, PASS, STAGE>(a, i, live, in, ctr, n_samples, o);
}

// FORM: 0 = the general state, 1 = compact state into stage 1, 2 = compact state into stages >= 2, 3 = the MIS integrators'
// state into stage 1 (see PathState)
template <bool MIS, int FORM>
EZD void shade_store(const WfArgs& a, const ShadeOut& o, const uint32_t k) {
  if (FORM == 1) {
    a.st_out.s0[k] = make_float4(o.f_r.x, o.f_r.y, o.f_r.z, o.cosine);
    reinterpret_cast<float2*>(a.st_out.s1)[k] = make_float2(__uint_as_float(o.sslot), __uint_as_float(o.tri0));
  } else if (FORM == 3) {
    a.st_out.s0[k] = make_float4(o.f_r.x, o.f_r.y, o.f_r.z, o.cosine);
    a.st_out.s1[k] = make_float4(o.pdf, __uint_as_float(o.sslot), __uint_as_float(o.seed), __uint_as_float(o.tri0));
  } else if (FORM == 2) {
    a.st_out.s0[k] = make_float4(o.history.x, o.history.y, o.history.z, o.cosine);
    a.st_out.s1[k] = make_float4(o.Lo.x, o.Lo.y, o.Lo.z, __uint_as_float(o.sslot));
    a.st_out.s2[k] = make_float4(o.f_r.x, o.f_r.y, o.f_r.z, __uint_as_float(o.tri0));
  } else {
    a.st_out.s0[k] = make_float4(o.history.x, o.history.y, o.history.z, o.cosine);
    a.st_out.s1[k] = make_float4(o.Lo.x, o.Lo.y, o.Lo.z, o.pdf);
    a.st_out.s2[k] = make_float4(o.f_r.x, o.f_r.y, o.f_r.z, __uint_as_float(o.sslot));
    a.st_out.s3[k] = make_float4(o.Le0.x, o.Le0.y, o.Le0.z, __uint_as_float(o.seed));
  }
  const bool shoot = !(o.flags & FLAG_TERMINATE);
  if (MIS) {
    a.st_out.s4[k] = make_float4(o.shadowC.x, o.shadowC.y, o.shadowC.z, __uint_as_float(o.flags));
    a.rq_out.o[k] = make_float4(o.P.x, o.P.y, o.P.z, 0.0f); // (one origin for the path's two rays: TraceQArgs::const_origin = 2)
    a.rq_out.d[2u * k] = make_float4(o.shadowL.x, o.shadowL.y, o.shadowL.z, (o.flags & FLAG_SHADOW_SHOT) ? 1.0f : 0.0f);
    a.rq_out.d[2u * k + 1u] = make_float4(o.rayL.x, o.rayL.y, o.rayL.z, shoot ? 1.0f : 0.0f);
  } else {
    a.rq_out.o[k] = make_float4(o.P.x, o.P.y, o.P.z, 0.0f);
    a.rq_out.d[k] = make_float4(o.rayL.x, o.rayL.y, o.rayL.z, 1.0f);
  }
}
template <int INTEG, int STAGE>
constexpr int store_form() { return compact_state<INTEG>() ? (STAGE == 0 ? 1 : 2) : ((integ_mis<INTEG>() && STAGE == 0) ? 3 : 0); }
// compaction of the surviving paths into the next queue: one ballot per wave, one atomic per workgroup
template <bool MIS, int FORM>
EZD void shade_emit(const WfArgs& a, const ShadeOut& o, uint32_t* alloc_lds) {
  const uint32_t k = block_alloc(a.n_out, o.emit, alloc_lds);
  if (!o.emit) return;
  shade_store<MIS, FORM>(a, o, k);
}

// Split shading (the two big stages of the timed route: ezrt_launch.hip launch_shade): most paths of a stage leave the scene, and everything a leaving path
// needs (state loads, environment lookup, sample store) fits in 45 registers -- so that part runs as
// its own kernel at twice the occupancy of the full shading kernel, and lists the paths with a surface
// interaction, which a second kernel then shades in dense waves.  The list is per workgroup (workgroup
// k of the second kernel takes the list of workgroup k of the first): no global atomic and no barrier
// in the first kernel -- one list-tail atomic per 512 paths was 32 k atomics on the primary stage, four
// times what the counter sustains in the time the kernel needs.
template <int INTEG, bool FULLCTR, int STAGE>
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_MISS_WAVES) void shade_miss_kernel(WfArgs a) {
  constexpr bool B0 = (STAGE == 0);
  __shared__ uint32_t list_tail;
  const uint32_t n_in = B0 ? a.n_slots : *a.n_in;
  const uint32_t stride = gridDim.x * SHADE_BLOCK;
  const uint32_t iters = (n_in + stride - 1) / stride;
  uint4* list = a.defer_list + (size_t)blockIdx.x * iters * SHADE_BLOCK; // this workgroup's region
  if (threadIdx.x == 0) list_tail = 0u;
  __syncthreads();
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t n_samples = 0;
  const int lane = threadIdx.x & 63;
  for (uint32_t k = 0, i = blockIdx.x * SHADE_BLOCK + threadIdx.x; k < iters; k++, i += stride) {
    ShadeOut o;
    ShadeIn in;
    shade_load<INTEG, 1, STAGE>(a, i, i < n_in, in);
    const bool deferred = shade_body<INTEG, FULLCTR, 1, STAGE>(a, i, i < n_in, in, ctr, n_samples, o);
    const unsigned long long m = ballot(deferred);
    if (m) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&list_tail, (uint32_t)__popcll(m)); // LDS
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (deferred) list[base + lane_rank(m)] = make_uint4(i, (uint32_t)in.h.x, (uint32_t)in.h.y, (uint32_t)in.sh.x);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.defer_count[blockIdx.x] = list_tail;
  if (B0) {
    unsigned long long sn = wave_sum(n_samples);
    if (lane == 0 && sn) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_SAMPLES], sn);
  }
  if (FULLCTR) {
    unsigned long long v5 = wave_sum(ctr.envmap), v6 = wave_sum(ctr.envcache);
    if (lane == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_MAP], v5);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_CACHE], v6);
    }
  }
}

// launched with the grid of shade_miss_kernel
template <int INTEG, bool FULLCTR, int STAGE>
// (The primary stage of the MIS integrators is compiled for 6 waves/SIMD, i.e. <= 80 VGPRs: with 8-wave workgroups the 83
// registers it would take mean two workgroups per CU, 80 mean three -- C4 +2 %.  Their later stages want 116-122 and
// would spill 50-100 registers; integrator 50 (81 VGPRs, two workgroups per CU) measures the same at 6 and at 7 waves --
// 76 and 72 registers, three workgroups: C2 and C3 within 0.3 % (round 3) -- and is left at 4.)
__global__ __launch_bounds__(SHADE_BLOCK, (integ_mis<INTEG>() && STAGE == 0) ? 6 : 4) void shade_hit_kernel(WfArgs a) {
  __shared__ uint32_t alloc_lds[SHADE_BLOCK / 64 + 1];
  constexpr bool B0 = (STAGE == 0);
  constexpr int FORM = store_form<INTEG, STAGE>();
  constexpr bool MIS = integ_mis<INTEG>();
  const uint32_t n_in = B0 ? a.n_slots : *a.n_in;
  const uint32_t stride = gridDim.x * SHADE_BLOCK;
  const uint32_t iters = (n_in + stride - 1) / stride;
  const uint4* list = a.defer_list + (size_t)blockIdx.x * iters * SHADE_BLOCK;
  const uint32_t cnt = a.defer_count[blockIdx.x];
  auto shade_entry = [&](const uint4 e, Counters& c, uint32_t& ns, ShadeOut& o) {
    ShadeIn in;
    shade_load<INTEG, 2, STAGE>(a, e.x, true, in, &e);
    shade_body<INTEG, FULLCTR, 2, STAGE>(a, e.x, true, in, c, ns, o);
  };
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t n_samples = 0;
  if (!MIS) {
    // Without MIS every surface interaction below the last bounce continues its path (P5/fsh:767-804 has
    // no early exit after a hit), so the workgroup reserves its cnt queue slots with ONE atomic and path j
    // of the list goes to slot base + j: no compaction, no barrier in the loop.
    if (cnt == 0u) return;
    if (threadIdx.x == 0) alloc_lds[0] = (a.bounce < a.p.max_bounce) ? atomicAdd(a.n_out, cnt) : 0u;
    __syncthreads();
    const uint32_t base = alloc_lds[0];
    for (uint32_t j = threadIdx.x; j < cnt; j += SHADE_BLOCK) {
      ShadeOut o;
      shade_entry(list[j], ctr, n_samples, o);
      if (o.emit) shade_store<MIS, FORM>(a, o, base + j);
      else if (a.bounce < a.p.max_bounce) // (a path that was pending and turned out to leave the scene: no ray in its slot)
        a.rq_out.d[base + j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  } else {
    for (uint32_t j0 = 0; j0 < cnt; j0 += SHADE_BLOCK) { // (workgroup-uniform trip count: barriers inside)
      const uint32_t j = j0 + threadIdx.x;
      ShadeOut o;
      o.emit = false;
      if (j < cnt) shade_entry(list[j], ctr, n_samples, o);
      shade_emit<MIS, FORM>(a, o, alloc_lds);
    }
  }
  if (FULLCTR) {
    const int lane = threadIdx.x & 63;
    unsigned long long v5 = wave_sum(ctr.envmap), v6 = wave_sum(ctr.envcache);
    if (lane == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_MAP], v5);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_CACHE], v6);
    }
  }
}

template <int INTEG, bool FULLCTR, int STAGE>
__global__ __launch_bounds__(SHADE_BLOCK, 4) void shade_kernel(WfArgs a) {
  __shared__ uint32_t alloc_lds[SHADE_BLOCK / 64 + 1];
  __shared__ uint32_t defer_list[SHADE_BLOCK];
  constexpr bool MIS = integ_mis<INTEG>();
  constexpr int FORM = store_form<INTEG, STAGE>();
  const uint32_t n_in = (STAGE == 0) ? a.n_slots : *a.n_in;
  const uint32_t stride = gridDim.x * SHADE_BLOCK;
  const uint32_t n_round = (n_in + stride - 1) / stride * stride; // keep workgroups whole for the barriers
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t n_samples = 0;

  for (uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x; i < n_round; i += stride) {
    ShadeOut o;
    if (STAGE == 0) {
      shade_path<INTEG, FULLCTR, 0, 0>(a, i, i < n_in, ctr, n_samples, o);
      shade_emit<MIS, FORM>(a, o, alloc_lds);
    } else {
      const bool deferred = shade_path<INTEG, FULLCTR, 1, STAGE>(a, i, i < n_in, ctr, n_samples, o);
      uint32_t n_def = 0;
      const uint32_t k = block_rank(deferred, alloc_lds, n_def);
      if (deferred) defer_list[k] = i;
      __syncthreads();
      const bool has = threadIdx.x < n_def; // (n_def <= SHADE_BLOCK: one dense round)
      const uint32_t j = has ? defer_list[threadIdx.x] : 0u;
      o.emit = false;
      if (has) shade_path<INTEG, FULLCTR, 2, STAGE>(a, j, true, ctr, n_samples, o);
      shade_emit<MIS, FORM>(a, o, alloc_lds);
    }
  }

  const int lane = threadIdx.x & 63;
  unsigned long long s = wave_sum(n_samples);
  if (lane == 0 && s) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_SAMPLES], s);
  if (FULLCTR) {
    unsigned long long v5 = wave_sum(ctr.envmap), v6 = wave_sum(ctr.envcache);
    if (lane == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_MAP], v5);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_CACHE], v6);
    }
  }
}

// ---------------------------------------------------------------------------
// Audit of the TIMED kernels (audit_via_queue): what traceq_kernel left in the hit records of one
// stage, written to ezrt_render_paths' per-pixel log (slot 0 = primary; bounce i: 1 + 2i = env shadow
// ray, 2 + 2i = bounce ray; -2 = ray not shot), and the chunk's sample radiance per pixel.
struct PathLogArgs {
  const int2* hits;      // hit records of the stage's ray queue
  const float4* rq_d;    // its ray directions (.w = 0: slot not shot)
  const float4* st_slot; // the path-state array that holds bits(sample slot) for this stage, and in which component
  int32_t slot_comp;     //   (general state: s2.w; compact state of integrator 50: float2 s1[i].x into stage 1, s1.w later)
  int32_t slot_stride;   //   floats per path in that array (4; 2 for the compact state into stage 1)
  const uint32_t* n_in;  // paths in the queue (stage >= 1)
  uint32_t n_slots;      // stage 0: pixel-samples of the chunk
  int32_t bounce;        // stage index b (0 = primary rays)
  int32_t mis;           // 1: two rays per path (2i = shadow, 2i + 1 = bounce)
  const int2* blocks;
  int32_t n_blocks;
  FastDiv div_blocks, div_sub;
  uint32_t frame_first;
  uint32_t scatter, scatter_shift;
  int32_t width, log_slots;
  EzrtRenderParams p;    // (pixel ownership of stage 0)
  int32_t* log_tri;
  float* log_t;
  float* log_colour;     // pathcolour_kernel
  const Sample3* samples;
};
__global__ __launch_bounds__(BLOCK) void pathlog_kernel(PathLogArgs a) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t n = a.bounce == 0 ? a.n_slots : *a.n_in;
  if (i >= n) return;
  const uint32_t sslot = a.bounce == 0 ? queue_to_sample(i, a.div_sub, a.scatter, a.scatter_shift)
                                       : __float_as_uint(reinterpret_cast<const float*>(a.st_slot)[(size_t)i * a.slot_stride + a.slot_comp]);
  int x, y;
  uint32_t frame;
  slot_to_pixel(a.blocks, a.div_blocks, sslot, a.frame_first, x, y, frame);
  if (frame != a.frame_first) return; // the log holds one frame
  const size_t pix = (size_t)y * a.width + x;
  auto put = [&](int slot, uint32_t r) {
    const int2 h = a.hits[r];
    a.log_tri[pix * a.log_slots + slot] = h.x < 0 ? -1 : h.x;
    a.log_t[pix * a.log_slots + slot] = h.x >= 0 ? __int_as_float(h.y) : INF;
  };
  if (a.bounce == 0) {
    if (!pixel_owned(a.p, x, y)) return; // pixel not owned by this shard: the caller's values stay
    for (int k = 0; k < a.log_slots; k++) {
      a.log_tri[pix * a.log_slots + k] = -2;
      a.log_t[pix * a.log_slots + k] = INF;
    }
    put(0, i);
  } else if (a.mis) {
    if (a.rq_d[2u * i].w != 0.0f) put(2 * a.bounce - 1, 2u * i);
    if (a.rq_d[2u * i + 1u].w != 0.0f) put(2 * a.bounce, 2u * i + 1u);
  } else {
    put(2 * a.bounce, i);
  }
}
__global__ __launch_bounds__(BLOCK) void pathcolour_kernel(PathLogArgs a, EzrtRenderParams p) {
  const uint32_t sslot = blockIdx.x * BLOCK + threadIdx.x; // first frame of the chunk
  if (sslot >= (uint32_t)a.n_blocks * 256u) return;
  int x, y;
  uint32_t frame;
  slot_to_pixel(a.blocks, a.div_blocks, sslot, a.frame_first, x, y, frame);
  if (!pixel_owned(p, x, y)) return;
  const Sample3 c = a.samples[sslot];
  const size_t pix = (size_t)y * a.width + x;
  a.log_colour[pix * 3 + 0] = c.x;
  a.log_colour[pix * 3 + 1] = c.y;
  a.log_colour[pix * 3 + 2] = c.z;
}

// ezrt_query_hits through traceq_kernel: caller rays -> a ray queue, hit records -> {tri, t}
__global__ void query_pack_kernel(const float* rays, uint32_t n, float4* o, float4* d, uint32_t* n_paths) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_paths = n;
  if (i >= n) return;
  const float* r = rays + (size_t)i * 6;
  o[i] = make_float4(r[0], r[1], r[2], 0.0f);
  d[i] = make_float4(r[3], r[4], r[5], 1.0f);
}
__global__ void query_unpack_kernel(const int2* hits, uint32_t n, int32_t* tri, float* t) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int2 h = hits[i];
  tri[i] = h.x < 0 ? -1 : h.x;
  t[i] = h.x >= 0 ? __int_as_float(h.y) : INF;
}

} // namespace ezd
