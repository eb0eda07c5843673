// ezrt_kernels.h -- the gfx950 kernels of the trace path.
//
//   trace_kernel<INTEG, FULLCTR, PATHLOG>   one thread = one pixel-sample
//       (ray-gen + hitBVH x (1 + k*bounces) + Disney BRDF + env lookups);
//       a 256-thread workgroup owns a 16x16 pixel block of one frame, each of
//       its four wavefronts an 8x8 sub-tile, so the 64 primary rays of a wave
//       are coherent.  Sample radiance goes to a frame-major sample buffer.
//   accumulate_kernel    the reference's running mean mix(last, c, 1/(k+1))
//       (P5/fsh:943-944) applied in frame order, one thread per pixel.
//   sobol_kernel / tonemap_kernel / query_kernel / math_kernel  small entry
//       points of the C ABI (KATs, pass3, probe rays, det-math audit).
//
// Grid shape: pixel-samples are independent, so the whole chunk of frames is
// one launch of n_blocks * n_frames workgroups (>> 256 CUs); the hardware
// dispatcher load-balances the wildly uneven per-pixel cost (sky pixel = 1 ray,
// bunny pixel = 5).  Block b lands on XCD b % 8 and blockIdx = frame * n_blocks
// + block, so with n_blocks % 8 == 0 one image block stays on one XCD's L2 for
// every frame of the chunk.
#pragma once
#include "ezrt_device.h"
#include "ezrt_records.h"

namespace ezd {

// dims 0-7: the shader literal (P5/fsh:351-353); dims 8-15: include/ezrt.h, ezrt_scene_set_sampler
__constant__ uint32_t c_sobol_v[16 * 32] = {
#include "ezrt_sobol_v.inc"
#include "ezrt_sobol_v16.inc"
};

// sobol(d, i): P5/fsh:361-369
EZD float sobol(uint32_t d, uint32_t i) {
  uint32_t result = 0, offset = d * 32u;
  for (uint32_t j = 0; i != 0; i >>= 1, j++)
    if (i & 1u) result ^= c_sobol_v[j + offset];
  return (float)result * (1.0f / (float)0xFFFFFFFFu);
}
EZD uint32_t gray_code(uint32_t i) { return i ^ (i >> 1); }

struct TraceArgs {
  DevScene sc;
  EzrtRenderParams p;
  const int2* blocks;   // origin (x, y) of each 16x16 pixel block to render
  int32_t n_blocks;
  uint32_t frame_first; // first frame of this launch
  Sample3* samples;     // [n_frames][n_blocks * 256]
  unsigned long long* counters; // EZRT_CTR_COUNT
  int32_t* log_tri;     // PATHLOG: [H][W][slots]
  float* log_t;
  float* log_colour;    // PATHLOG: [H][W][3]
  int32_t stack_entries;
};

EZD bool pixel_owned(const EzrtRenderParams& p, int x, int y) {
  if (x < p.x0 || x >= p.x1 || y < p.y0 || y >= p.y1) return false;
  if (p.shard_count <= 1) return true;
  int tw = p.tile_w > 0 ? p.tile_w : 32, th = p.tile_h > 0 ? p.tile_h : 32;
  int tiles_x = (p.width + tw - 1) / tw;
  int tile = (y / th) * tiles_x + (x / tw);
  return tile % p.shard_count == p.shard_index;
}

EZD unsigned long long wave_sum(uint32_t v) {
  unsigned long long s = v;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

// ---- queue position -> pixel-sample -> primary ray (used by raygen_kernel, and by the primary stage's trace and shading
// kernels when the chunk's rays are generated where they are consumed: Tuning::gen_primary)
EZD void slot_to_pixel(const int2* blocks, const FastDiv& n_blocks, uint32_t slot, uint32_t frame_first, int& x, int& y,
                       uint32_t& frame) {
  uint32_t tid = slot & 255u;
  uint32_t b = slot >> 8;
  const uint32_t fk = fastdiv(b, n_blocks), blk = b - fk * n_blocks.d;
  int2 org = blocks[blk];
  uint32_t wave = tid >> 6, lane = tid & 63u;
  x = org.x + (int)((wave & 1u) * 8u + (lane & 7u));
  y = org.y + (int)((wave >> 1) * 8u + (lane >> 3));
  frame = frame_first + fk;
}

// Queue order of the primary rays.  Sample slots are laid out [frame][16x16 block][8x8 sub-block]
// [lane]; taking queue position = sample slot would hand a wave pools of 256 rays from ONE 16x16
// pixel block -- all cheap (sky) or all expensive (the Bunny), and the launch ends with the waves
// that drew the expensive ones.  So the 8x8 sub-blocks of a frame are visited in a scattered
// order, sub-block (r * scatter) mod n_sub at position r (scatter ~ 2531, coprime to n_sub): a pool is four
// sub-blocks from distant parts of the image and every pool costs about the same.
EZD uint32_t queue_to_sample(uint32_t qslot, const FastDiv& n_sub_div, uint32_t scatter, uint32_t sh = 6u) {
  const uint32_t n_sub = n_sub_div.d; // granules of 1 << sh slots per frame = (n_blocks * 256) >> sh
  const uint32_t q = qslot >> sh;
  const uint32_t fk = fastdiv(q, n_sub_div), r = q - fk * n_sub;
  const uint32_t p = r * scatter; // 32-bit: the host keeps n_sub <= 2^20 and scatter < 2^12 here
  const uint32_t r2 = p - fastdiv(p, n_sub_div) * n_sub;
  return ((fk * n_sub + r2) << sh) | (qslot & ((1u << sh) - 1u));
}

// The primary ray of queue position `qslot`: main() up to the hitBVH call (P5/fsh:315-318 seed, 920-925 jitter, camera,
// normalize) -- (dir.xyz, 1), or w = 0 for a pixel this shard does not own.  ONE definition for every kernel that needs the
// direction, so that the ray the trace follows and the ray the shading stage shades are the same bits.
EZD float4 primary_dir(const EzrtRenderParams& p, const int2* blocks, const FastDiv& div_blocks, const FastDiv& div_sub, uint32_t scatter,
                       uint32_t scatter_shift, uint32_t frame_first, uint32_t qslot) {
  int x, y;
  uint32_t frame;
  slot_to_pixel(blocks, div_blocks, queue_to_sample(qslot, div_sub, scatter, scatter_shift), frame_first, x, y, frame);
  if (!pixel_owned(p, x, y)) return make_float4(0, 0, 0, 0.0f);
  const uint32_t ix = (uint32_t)x, iy = (uint32_t)y;
  uint32_t seed = (ix * 1973u + iy * 9277u + frame * 26699u) | 1u;
  const float W = (float)p.width, H = (float)p.height;
  // (x / W for a power-of-two W IS x * (1 / W) on the bits, and every BASELINE frame is one: the four divisions below as
  // multiplications behind a uniform branch were built and measured in round 4 -- C2 -3.3 %: four more launch-invariant values
  // in a kernel at its SGPR limit became three more VGPR spills in the refill block; profiles/r4/rcp_pow2_ab.txt)
  float pixx = ((float)ix + 0.5f) / W * 2.0f - 1.0f;
  float pixy = ((float)iy + 0.5f) / H * 2.0f - 1.0f;
  float aax = (rnd(seed) - 0.5f) / W;
  float aay = (rnd(seed) - 0.5f) / H;
  float vx = pixx + aax, vy = pixy + aay, vz = -1.5f;
  const float* m = p.camera_rotate;
  f3 dir = mk(m[0] * vx + m[4] * vy + m[8] * vz, m[1] * vx + m[5] * vy + m[9] * vz, m[2] * vx + m[6] * vy + m[10] * vz);
  dir = normalize(dir);
  return make_float4(dir.x, dir.y, dir.z, 1.0f);
}

template <bool PATHLOG>
EZD void plog(const TraceArgs& a, size_t pix, int slot, int32_t tri, float t) {
  if (PATHLOG) {
    int slots = 1 + 2 * a.p.max_bounce;
    a.log_tri[pix * slots + slot] = tri;
    a.log_t[pix * slots + slot] = (tri >= 0) ? t : INF;
  }
}

template <int INTEG, bool FULLCTR, bool PATHLOG>
__global__ __launch_bounds__(BLOCK) void trace_kernel(TraceArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_stack[];
  constexpr bool P5TRI = (INTEG >= 50);
  constexpr bool MIS = integ_mis<INTEG>();
  constexpr bool ANISO_IS = integ_aniso_is<INTEG>();
  const int tid = threadIdx.x;
  const int blk = blockIdx.x % a.n_blocks;
  const int fk = blockIdx.x / a.n_blocks;
  const uint32_t frame = a.frame_first + (uint32_t)fk;
  const int wave = tid >> 6, lane = tid & 63;
  const int2 org = a.blocks[blk];
  const int x = org.x + (wave & 1) * 8 + (lane & 7);
  const int y = org.y + (wave >> 1) * 8 + (lane >> 3);
  int* stack = lds_stack + tid;
  const DevScene& sc = a.sc;
  const EzrtRenderParams& p = a.p;

  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  uint32_t samples = 0;
  f3 colour = mk(0, 0, 0);
  const bool active = pixel_owned(p, x, y);
  if (active) {
    samples = 1;
    const size_t pix = (size_t)y * p.width + x;
    if (PATHLOG) {
      int slots = 1 + 2 * p.max_bounce;
      for (int k = 0; k < slots; k++) {
        a.log_tri[pix * slots + k] = -2;
        a.log_t[pix * slots + k] = INF;
      }
    }
    // main(): P5/fsh:315-318, 920-925
    const uint32_t ix = (uint32_t)x, iy = (uint32_t)y;
    uint32_t seed = (ix * 1973u + iy * 9277u + frame * 26699u) | 1u;
    const float W = (float)p.width, H = (float)p.height;
    float pixx = ((float)ix + 0.5f) / W * 2.0f - 1.0f;
    float pixy = ((float)iy + 0.5f) / H * 2.0f - 1.0f;
    float aax = (rnd(seed) - 0.5f) / W;
    float aay = (rnd(seed) - 0.5f) / H;
    float vx = pixx + aax, vy = pixy + aay, vz = -1.5f;
    const float* m = p.camera_rotate;
    f3 dir = mk(m[0] * vx + m[4] * vy + m[8] * vz, m[1] * vx + m[5] * vy + m[9] * vz,
                m[2] * vx + m[6] * vy + m[10] * vz);
    dir = normalize(dir);
    f3 org3 = mk(p.eye[0], p.eye[1], p.eye[2]);

    int32_t tri;
    float t;
    hit_bvh<FULLCTR, BLOCK>(sc, org3, dir, stack, tri, t, ctr);
    plog<PATHLOG>(a, pix, 0, tri, t);
    if (tri < 0) {
      colour = hdr_color<FULLCTR>(sc, dir, p.env_clamp, ctr);
    } else {
      Hit hit;
      shade_point<P5TRI>(sc, tri, t, org3, dir, hit);
      const f3 Le0 = hit.m.emissive;
      f3 Lo = mk(0, 0, 0), history = mk(1, 1, 1);
      float cpu = 0.0f, cpv = 0.0f;
      if (INTEG >= 50) cp_offsets(ix, iy, cpu, cpv);
      const uint32_t gray = gray_code(frame + 1u);

      for (int bounce = 0; bounce < p.max_bounce; bounce++) {
        const f3 V = -hit.viewDir, N = hit.N;
        f3 X = mk(0, 0, 0), Y = mk(0, 0, 0);
        if (ANISO_IS) get_tangent(N, X, Y);
        if (MIS) {
          // env importance sample + shadow ray: P5/fsh:819-842
          float h1 = rnd(seed);
          float h2 = rnd(seed);
          f3 Lh = sample_hdr<FULLCTR>(sc, h1, h2, ctr);
          if (dot(N, Lh) > 0.0f) {
            int32_t st;
            float stt;
            hit_bvh<FULLCTR, BLOCK>(sc, hit.P, Lh, stack, st, stt, ctr);
            plog<PATHLOG>(a, pix, 1 + 2 * bounce, st, stt);
            if (st < 0) {
              f3 color;
              float pdf_light;
              hdr_color_pdf<FULLCTR>(sc, Lh, p.env_clamp, ctr, color, pdf_light);
              f3 f_r;
              float pdf_brdf;
              brdf_evaluate_pdf<ANISO_IS>(V, N, Lh, X, Y, hit.m, f_r, pdf_brdf);
              float w = mis_mix_weight(pdf_light, pdf_brdf);
              Lo = Lo + (((history * w) * color) * f_r) * dot(N, Lh) / pdf_light;
            }
          }
        }
        // sample direction
        f3 L;
        float xi1, xi2;
        if (INTEG >= 50) { // sobolVec2 + CP: P5/fsh:771-772, 845-846 (dims wrap at 8)
          uint32_t d0 = ((uint32_t)bounce * 2u) & sc.sobol_mask, d1 = ((uint32_t)bounce * 2u + 1u) & sc.sobol_mask;
          xi1 = cp_rotate(sobol(d0, gray), cpu);
          xi2 = cp_rotate(sobol(d1, gray), cpv);
        } else { // P3/fsh:109-114: z = rand() then phi = 2 pi rand()
          xi1 = rnd(seed);
          xi2 = rnd(seed);
        }
        float cosine, pdf;
        f3 f_r;
        if (MIS) {
          float xi3 = rnd(seed);
          L = ANISO_IS ? sample_brdf_aniso(xi1, xi2, xi3, V, N, X, Y, hit.m) : sample_brdf(xi1, xi2, xi3, V, N, hit.m);
          cosine = dot(N, L);
          if (cosine <= 0.0f) break;
        } else {
          L = to_normal_hemisphere(sample_hemisphere(xi1, xi2), N);
          pdf = 1.0f / (2.0f * PI);
          cosine = ez_max(0.0f, dot(L, N));
          if (INTEG == EZRT_INTEGRATOR_P3_DIFFUSE) {
            f_r = hit.m.baseColor / PI;
          } else {
            f3 tangent, bitangent;
            get_tangent(N, tangent, bitangent);
            f_r = brdf_evaluate<INTEG == EZRT_INTEGRATOR_P4_DISNEY>(V, N, L, tangent, bitangent, hit.m);
          }
        }
        int32_t nt;
        float ntt;
        hit_bvh<FULLCTR, BLOCK>(sc, hit.P, L, stack, nt, ntt, ctr);
        plog<PATHLOG>(a, pix, 2 + 2 * bounce, nt, ntt);
        if (MIS) {
          brdf_evaluate_pdf<ANISO_IS>(V, N, L, X, Y, hit.m, f_r, pdf);
          if (pdf <= 0.0f) break;
        }
        if (nt < 0) {
          f3 sky;
          float pdf_light = 0.0f;
          if (MIS) hdr_color_pdf<FULLCTR>(sc, L, p.env_clamp, ctr, sky, pdf_light);
          else sky = hdr_color<FULLCTR>(sc, L, p.env_clamp, ctr);
          if (MIS) {
            float w = mis_mix_weight(pdf, pdf_light);
            Lo = Lo + (((history * w) * sky) * f_r) * cosine / pdf;
          } else {
            Lo = Lo + ((history * sky) * f_r) * cosine / pdf;
          }
          break;
        }
        Hit nh;
        shade_point<P5TRI>(sc, nt, ntt, hit.P, L, nh);
        Lo = Lo + ((history * nh.m.emissive) * f_r) * cosine / pdf;
        history = history * (f_r * cosine / pdf);
        hit = nh;
      }
      colour = Le0 + Lo;
    }
    if (PATHLOG) {
      a.log_colour[pix * 3 + 0] = colour.x;
      a.log_colour[pix * 3 + 1] = colour.y;
      a.log_colour[pix * 3 + 2] = colour.z;
    }
  }
  if (!PATHLOG) a.samples[((size_t)fk * a.n_blocks + blk) * BLOCK + tid] = Sample3{colour.x, colour.y, colour.z};

  // counters: one atomic per wave per slot
  unsigned long long r = wave_sum(ctr.rays), s = wave_sum(samples);
  if (lane == 0) {
    atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_RAYS], r);
    atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_SAMPLES], s);
  }
  if (FULLCTR) {
    unsigned long long v1 = wave_sum(ctr.pops), v2 = wave_sum(ctr.inner), v3 = wave_sum(ctr.tris),
                       v4 = wave_sum(ctr.mats), v5 = wave_sum(ctr.envmap), v6 = wave_sum(ctr.envcache);
    if (lane == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_NODE_POPS], v1);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_INNER_POPS], v2);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_TRI_TESTS], v3);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_MAT_FETCH], v4);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_MAP], v5);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_ENV_CACHE], v6);
    }
  }
}

// mix(lastColor, color, 1.0/float(frameCounter+1)): P5/fsh:943-947, in frame order.
struct AccumArgs {
  EzrtRenderParams p;
  const int2* blocks;
  int32_t n_blocks;
  uint32_t frame_first, n_frames;
  const Sample3* samples;
  float4* accum; // [H][W] RGBA
};
__global__ __launch_bounds__(BLOCK) void accumulate_kernel(AccumArgs a) {
  const int tid = threadIdx.x, blk = blockIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int2 org = a.blocks[blk];
  const int x = org.x + (wave & 1) * 8 + (lane & 7);
  const int y = org.y + (wave >> 1) * 8 + (lane >> 3);
  if (!pixel_owned(a.p, x, y)) return;
  const size_t pix = (size_t)y * a.p.width + x;
  float4 last = a.accum[pix];
  f3 mean = mk(last.x, last.y, last.z);
  for (uint32_t k = 0; k < a.n_frames; k++) {
    const Sample3 c = a.samples[((size_t)k * a.n_blocks + blk) * BLOCK + tid];
    uint32_t frame = a.frame_first + k;
    if (frame == 0) {
      mean = mk(c.x, c.y, c.z);
    } else {
      float w = 1.0f / (float)(frame + 1u);
      mean = mix3(mean, mk(c.x, c.y, c.z), w);
    }
  }
  a.accum[pix] = make_float4(mean.x, mean.y, mean.z, 1.0f);
}

// inner records with both child boxes translated by -S (S = the eye): what hitAABB subtracts on every visit
__global__ void inner_rel_kernel(const float4* inner, int n_inner, float sx, float sy, float sz, float4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  const float4 q0 = inner[i * 4], q1 = inner[i * 4 + 1], q2 = inner[i * 4 + 2];
  out[i * 4] = make_float4(q0.x - sx, q0.y - sy, q0.z - sz, q0.w - sx);
  out[i * 4 + 1] = make_float4(q1.x - sy, q1.y - sz, q1.z - sx, q1.w - sy);
  out[i * 4 + 2] = make_float4(q2.x - sz, q2.y - sx, q2.z - sy, q2.w - sz);
  out[i * 4 + 3] = inner[i * 4 + 3];
}

__global__ void sobol_kernel(uint32_t index0, int n, int n_dims, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n_dims) return;
  int s = i / n_dims, d = i % n_dims;
  out[i] = sobol((uint32_t)d, gray_code(index0 + (uint32_t)s));
}

// ezrt_frame_nonfinite: pixels whose running mean is poisoned (the reference's 0/0 in misMixWeight, P5/fsh:754-757, reproduced:
// include/ezrt.h "Numerical contract").  One ballot per wave, one atomic per wave that saw any.
__global__ void nonfinite_kernel(const float4* rgba, size_t n, unsigned long long* count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((n + 63) & ~(size_t)63); i += (size_t)gridDim.x * blockDim.x) {
    bool bad = false;
    if (i < n) {
      const float4 v = rgba[i];
      // non-finite = exponent all ones
      bad = ((__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u) ||
            ((__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u);
    }
    const unsigned long long m = __ballot(bad);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(count, (unsigned long long)__popcll(m));
  }
}

// pass3.fsh:14-24 + P1/main.cpp:187-189
__global__ void tonemap_kernel(const float4* rgba, int n, uint8_t* rgb8) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 c = rgba[i];
  float lum = 0.3f * c.x + 0.6f * c.y + 0.1f * c.z;
  float k = 1.0f + lum / 1.5f;
  float ch[3] = {c.x, c.y, c.z};
  for (int j = 0; j < 3; j++) {
    float v = ch[j] * 1.0f / k;
    v = ez_pow(v, 1.0f / 2.2f);
    float q = ez_clamp(v * 255.0f, 0.0f, 255.0f);
    if (!(q == q)) q = 0.0f;
    rgb8[(size_t)i * 3 + j] = (uint8_t)(int)q;
  }
}

struct QueryArgs {
  DevScene sc;
  const float* rays;
  int n;
  int32_t* tri;
  float* t;
  unsigned long long* counters;
};
template <bool FULLCTR>
__global__ __launch_bounds__(BLOCK) void query_kernel(QueryArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_stack[];
  int i = blockIdx.x * BLOCK + threadIdx.x;
  Counters ctr = {0, 0, 0, 0, 0, 0, 0};
  if (i < a.n) {
    const float* r = a.rays + (size_t)i * 6;
    int32_t tri;
    float t;
    hit_bvh<FULLCTR, BLOCK>(a.sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), lds_stack + threadIdx.x, tri, t, ctr);
    a.tri[i] = tri;
    a.t[i] = (tri >= 0) ? t : INF;
  }
  unsigned long long rr = wave_sum(ctr.rays);
  if ((threadIdx.x & 63) == 0) atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_RAYS], rr);
  if (FULLCTR) {
    unsigned long long v1 = wave_sum(ctr.pops), v2 = wave_sum(ctr.inner), v3 = wave_sum(ctr.tris), v4 = wave_sum(ctr.mats);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_NODE_POPS], v1);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_INNER_POPS], v2);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_TRI_TESTS], v3);
      atomicAdd(&ctr_slot(a.counters)[EZRT_CTR_MAT_FETCH], v4);
    }
  }
}

// ezrt_debug_math op 17: the launch-invariant division of the queue maps (FastDiv), bits in / bits out
__global__ void fastdiv_kernel(const float* a, FastDiv f, int n, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = __uint_as_float(fastdiv(__float_as_uint(a[i]), f));
}

__global__ void math_kernel(int op, const float* a, const float* b, int n, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = a[i], y = b[i], r;
  switch (op) {
    case 0: r = ez_sin(x); break;
    case 1: r = ez_cos(x); break;
    case 2: r = ez_atan2(x, y); break;
    case 3: r = ez_asin(x); break;
    case 4: r = ez_log(x); break;
    case 5: r = ez_exp(x); break;
    case 6: r = ez_pow(x, y); break;
    case 7: r = __builtin_sqrtf(x); break;
    case 8: r = x / y; break;
    default: {
      uint32_t sd = __float_as_uint(x);
      r = rnd(sd);
    }
  }
  out[i] = r;
}

// Intersector audit (ezrt_debug_math ops 10-12): the device functions the traversal kernels call,
// evaluated on caller-supplied operands so tests can compare them with the reference's C++ twins
// (P2/main.cpp:212-238, 449-463) directly.  op 10 = hit_aabb (exact select form), 12 = hit_aabb_tame
// (v_min3/v_max3 form; NaN for rays that are not tame = the caller must not have used it), 11 =
// hit_triangle_t on the 48-B record ezrt_scene_create builds (unit plane normal precomputed with the
// same fp32 operations), INF on a miss.
__global__ void isect_kernel(int op, const float* a, const float* b, int n, float* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = a + 6 * (size_t)i;
  f3 S = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
  if (op >= 13) { // integrator 52's sampler / pdf in the frame N = (0,0,1) (include/ezrt.h, ezrt_debug_math)
    const float* q = b + 6 * (size_t)i;
    Mat m;
    m.emissive = mk(0, 0, 0);
    m.baseColor = mk(0, 0, 0);
    m.subsurface = m.specular = m.specularTint = m.sheen = m.sheenTint = 0.0f;
    m.roughness = q[0];
    m.anisotropic = q[1];
    m.metallic = q[2];
    m.clearcoat = q[3];
    m.clearcoatGloss = q[4];
    m.anisotropic = q[1];
    mat_derive(m);
    const f3 N = mk(0, 0, 1);
    f3 X, Y;
    get_tangent(N, X, Y);
    if (op == 13) {
      out[i] = brdf_pdf_aniso(S, N, d, X, Y, m);
    } else {
      const f3 L = sample_brdf_aniso(r[0], r[1], r[2], d, N, X, Y, m);
      out[i] = op == 14 ? L.x : (op == 15 ? L.y : L.z);
    }
    return;
  }
  if (op == 11) {
    const float* t = b + 9 * (size_t)i;
    float e1x = t[3] - t[0], e1y = t[4] - t[1], e1z = t[5] - t[2];
    float e2x = t[6] - t[0], e2y = t[7] - t[1], e2z = t[8] - t[2];
    float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
    float inv = 1.0f / __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
    float4 g[3] = {make_float4(t[0], t[1], t[2], cx * inv), make_float4(t[3], t[4], t[5], cy * inv),
                   make_float4(t[6], t[7], t[8], cz * inv)};
    float tt;
    out[i] = hit_triangle_t(g, S, d, tt) ? tt : INF;
    return;
  }
  const float* q = b + 6 * (size_t)i;
  f3 inv = mk(ez_rcp(d.x), ez_rcp(d.y), ez_rcp(d.z));
  f3 AA = mk(q[0], q[1], q[2]), BB = mk(q[3], q[4], q[5]);
  if (op == 10) out[i] = hit_aabb(S, inv, AA, BB);
  else out[i] = ray_is_tame(S, inv) ? hit_aabb_tame(S, inv, AA, BB) : __uint_as_float(0x7fc00000u);
}

// ezrt_debug_math op 18: ez_rcp(x) against the compiler's `1.0f / x` for ALL 2^32 bit patterns of x (a NaN equals a NaN).
// res[0] = mismatches, res[1] = the smallest mismatching pattern.
__global__ void rcp_audit_kernel(unsigned long long* res) {
  unsigned long long bad = 0ull, first = ~0ull;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32); k += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)k);
    float want;
    asm volatile("" : "=v"(want) : "0"(1.0f / x)); // (keep the two expressions apart)
    const float got = ez_rcp(x);
    const bool same = __float_as_uint(got) == __float_as_uint(want) || (got != got && want != want);
    if (!same) {
      bad++;
      if (k < first) first = k;
    }
  }
  if (bad) {
    atomicAdd(&res[0], bad);
    atomicMin(&res[1], first);
  }
}

} // namespace ezd
