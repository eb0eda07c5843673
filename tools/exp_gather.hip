// exp_gather.hip -- micro-benchmark behind DESIGN.md's "what bounds traceq_kernel" paragraph:
// how fast does one CU gather 64-byte records from an L2-resident table when
//   A  every lane fetches its own record with four 16-byte loads (what traceq_kernel does),
//   B  four adjacent lanes fetch one record together (one 16-byte load per lane),
//   C  every lane fetches 16 bytes of its own record (one load),
//   D  as A but the four loads of a lane go to four different records (no shared line).
// Each variant is a dependent chain (next index from the loaded data), 5 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_gather tools/exp_gather.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned mix(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const f4* __restrict__ table, unsigned mask, int iters, float* out) {
  unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned idx = mix(tid * 2654435761U + 12345U) & mask;
  float acc = 0.0f;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      const f4* r = table + (size_t)idx * 4;
      f4 a = r[0], b = r[1], c = r[2], d = r[3];
      float s = a.x + b.y + c.z + d.w;
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 1) {
      unsigned q = __shfl(idx, threadIdx.x & ~3u, 64); // quad leader's index
      f4 a = table[(size_t)q * 4 + (threadIdx.x & 3)];
      float s = a.x + a.y;
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 2) {
      f4 a = table[(size_t)idx * 4];
      float s = a.x + a.y;
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 4 || MODE == 5 || MODE == 6) { // G adjacent lanes (2, 8, 16) read consecutive 16-B pieces of a run starting at a random record
      constexpr unsigned G = MODE == 4 ? 2u : (MODE == 5 ? 8u : 16u);
      unsigned q = __shfl(idx, threadIdx.x & ~(G - 1u), 64);
      f4 a = table[((size_t)q * 4 + (threadIdx.x & (G - 1u))) & ((size_t)mask * 4 + 3)];
      float s = a.x + a.y;
      for (unsigned o = 1; o < G; o <<= 1) s += __shfl_xor(s, o, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 7) { // 3 loads per lane at stride 48 B inside groups of 8 lanes (a leaf's triangles, one per lane: today's g = 8 pattern)
      unsigned q = __shfl(idx, threadIdx.x & ~7u, 64);
      const size_t base = ((size_t)q * 4 + (threadIdx.x & 7u) * 3) & ((size_t)mask * 4 + 3);
      f4 a = table[base], b = table[(base + 1) & ((size_t)mask * 4 + 3)], c = table[(base + 2) & ((size_t)mask * 4 + 3)];
      float s = a.x + b.y + c.z;
      for (unsigned o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 8) { // the same 24 pieces read leaf-SoA: load p of lane j = piece p * 8 + j (8 lanes x 16 B contiguous per load)
      unsigned q = __shfl(idx, threadIdx.x & ~7u, 64);
      const size_t base = (size_t)q * 4 + (threadIdx.x & 7u), m4 = (size_t)mask * 4 + 3;
      f4 a = table[base & m4], b = table[(base + 8) & m4], c = table[(base + 16) & m4];
      float s = a.x + b.y + c.z;
      for (unsigned o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 9) { // pairs: 3 loads per lane at stride 48 B in groups of 2 (today's g = 2 pattern)
      unsigned q = __shfl(idx, threadIdx.x & ~1u, 64);
      const size_t base = ((size_t)q * 4 + (threadIdx.x & 1u) * 3), m4 = (size_t)mask * 4 + 3;
      f4 a = table[base & m4], b = table[(base + 1) & m4], c = table[(base + 2) & m4];
      float s = a.x + b.y + c.z;
      s += __shfl_xor(s, 1, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else if (MODE == 10) { // pairs, leaf-SoA with n = 6: piece p of triangle k at p * 6 + k
      unsigned q = __shfl(idx, threadIdx.x & ~1u, 64);
      const size_t base = ((size_t)q * 4 + (threadIdx.x & 1u)), m4 = (size_t)mask * 4 + 3;
      f4 a = table[base & m4], b = table[(base + 6) & m4], c = table[(base + 12) & m4];
      float s = a.x + b.y + c.z;
      s += __shfl_xor(s, 1, 64);
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    } else {
      unsigned i1 = mix(idx + 1) & mask, i2 = mix(idx + 2) & mask, i3 = mix(idx + 3) & mask;
      f4 a = table[(size_t)idx * 4], b = table[(size_t)i1 * 4 + 1], c = table[(size_t)i2 * 4 + 2],
         d = table[(size_t)i3 * 4 + 3];
      float s = a.x + b.y + c.z + d.w;
      acc += s;
      idx = mix(idx + __float_as_uint(s)) & mask;
    }
  }
  out[tid] = acc;
}

template <int MODE>
static void run(const char* name, const f4* table, unsigned mask, int iters, float* out, int blocks, double recs_per_thread_iter,
                double loads_per_thread_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, mask, 16, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, mask, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  double threads = (double)blocks * 256;
  double recs = threads * iters * recs_per_thread_iter;
  double loads = threads * iters * loads_per_thread_iter;
  double clk = ms * 1e-3 * 2.4e9 * 256; // CU-clocks
  printf("%-28s table %6u KB  %8.3f ms  %8.1f Mrec/ms  lane-loads/clk/CU %.3f  bytes/clk/CU %.2f  step %.0f clk\n", name,
         (mask + 1) * 64 / 1024, ms, recs / ms * 1e-6, loads / clk, loads * 16 / clk, ms * 1e-3 * 2.4e9 / iters);
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 4000;
  int blocks = 256 * 5;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  for (unsigned nrec : {1u << 7, 1u << 14, 1u << 17, 1u << 22}) {
    std::vector<float> h((size_t)nrec * 16);
    unsigned s = 1;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = (float)(s >> 8) * (1.0f / 16777216.0f);
    }
    f4* table;
    hipMalloc(&table, h.size() * 4);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("A lane-owned 4x16B", table, nrec - 1, iters, out, blocks, 1.0, 4.0);
    run<1>("B quad-shared 1x16B", table, nrec - 1, iters, out, blocks, 0.25, 1.0);
    run<2>("C lane-owned 1x16B", table, nrec - 1, iters, out, blocks, 1.0, 1.0);
    run<3>("D lane-owned 4 lines", table, nrec - 1, iters, out, blocks, 4.0, 4.0);
    run<4>("E pair-shared 1x16B", table, nrec - 1, iters, out, blocks, 0.5, 1.0);
    run<5>("F oct-shared 1x16B", table, nrec - 1, iters, out, blocks, 0.125, 1.0);
    run<6>("G 16-shared 1x16B", table, nrec - 1, iters, out, blocks, 0.0625, 1.0);
    run<7>("H 8 lanes x 3 @48B stride", table, nrec - 1, iters, out, blocks, 1.0, 3.0);
    run<8>("I 8 lanes x 3 leaf-SoA", table, nrec - 1, iters, out, blocks, 1.0, 3.0);
    run<9>("J 2 lanes x 3 @48B stride", table, nrec - 1, iters, out, blocks, 1.0, 3.0);
    run<10>("K 2 lanes x 3 leaf-SoA", table, nrec - 1, iters, out, blocks, 1.0, 3.0);
    hipFree(table);
  }
  return 0;
}
