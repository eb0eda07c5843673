// ezrt_records.h -- layout constants of the device records that BOTH the scene builder (ezrt_scene_build.hip) and the traversal kernels
// (ezrt_traceq.h, ezrt_traceq4.h) must agree on.  DESIGN.md 4.
#pragma once
#include <cstdint>

namespace ezd {

constexpr int BLOCK = 256; // threads per workgroup of the trace kernels (the LDS traversal stack is [row][BLOCK])

// 4-wide record (128 B in HBM = one L2 line, 112 B in LDS).  The near / far plane of each axis is SELECTED by the sign of 1/direction
// through the address of the 16-byte row that is loaded (ezrt_traceq4.h), so the BB row of an axis lies 64 bytes after its AA row:
//   AAx[4] AAy[4] AAz[4] ref[4] BBx[4] BBy[4] BBz[4] (pad)
constexpr int N4_ROW_AA = 0, N4_ROW_BB = 4, N4_ROW_REF = 3;
constexpr uint32_t REF_EMPTY = 0xfffffffdu; // unused slot of a 4-wide record
constexpr int N4_FLOAT4 = 8;                // record stride in HBM, float4s (7 used)
constexpr int N4_LDS_DWORDS = 28;           // record stride in LDS: 112 B; 28 r mod 64 hits 16 distinct bank quads
constexpr uint32_t REF_NOPRUNE = 0x40000000u; // inner reference (and root4): a triangle without a useful bound lies below this record
constexpr uint32_t REF_INDEX = 0x00ffffffu;   // ... its record index

} // namespace ezd
