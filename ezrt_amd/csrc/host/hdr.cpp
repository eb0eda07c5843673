// hdr.cpp -- Radiance .hdr (RGBE, RLE) reader and the env-map importance cache.
// Restates P5/lib/hdrloader.cpp:50-212 (Igor Kravtchenko's loader as vendored
// by the reference) and P5/main.cpp:592-689 (calculateHdrCache).  New code.
#include <cstdio>
#include <cstring>
#include <vector>

#include "ezrt_detmath.h"
#include "ezrt_scene.hpp"

namespace ezrt {

namespace {

struct Reader {
  const unsigned char* p;
  const unsigned char* e;
  bool eof = false;
  int get() {
    if (p >= e) {
      eof = true;
      return -1;
    }
    return *p++;
  }
  void unget() {
    --p;
  }
};

typedef unsigned char RGBE[4];

// oldDecrunch: hdrloader.cpp:182-212 (flat RGBE with optional old-style runs)
// `base` = first pixel of the scanline: a run copies the previous pixel, which exists from the second pixel of the
// LINE on (decrunch's fallback enters at scan + 1 with pixel 0 already written, hdrloader.cpp:135-139).
bool old_decrunch(RGBE* scan, int len, Reader& f, RGBE* base) {
  int rshift = 0;
  while (len > 0) {
    int r = f.get(), g = f.get(), b = f.get(), x = f.get();
    if (f.eof) return false;
    scan[0][0] = (unsigned char)r;
    scan[0][1] = (unsigned char)g;
    scan[0][2] = (unsigned char)b;
    scan[0][3] = (unsigned char)x;
    if (r == 1 && g == 1 && b == 1) {
      if (scan == base) return false; // a run needs a previous pixel (the reference reads before the buffer)
      if (rshift >= 24) return false; // four run markers in a row: `x << rshift` leaves int (undefined in the reference);
                                      // no scanline (len <= 0x7fff on this path, any len in practice) is that long
      for (int i = x << rshift; i > 0 && len > 0; i--) {
        memcpy(&scan[0][0], &scan[-1][0], 4);
        scan++;
        len--;
      }
      rshift += 8;
    } else {
      scan++;
      len--;
      rshift = 0;
    }
  }
  return true;
}

// decrunch: hdrloader.cpp:139-180 (new-style per-component RLE)
bool decrunch(RGBE* scan, int len, Reader& f) {
  if (len < 8 || len > 0x7fff) return old_decrunch(scan, len, f, scan);
  int i = f.get();
  if (i != 2) {
    f.unget();
    return old_decrunch(scan, len, f, scan);
  }
  scan[0][1] = (unsigned char)f.get();
  scan[0][2] = (unsigned char)f.get();
  i = f.get();
  if (scan[0][1] != 2 || (scan[0][2] & 128)) {
    scan[0][0] = 2;
    scan[0][3] = (unsigned char)i;
    return old_decrunch(scan + 1, len - 1, f, scan);
  }
  for (int comp = 0; comp < 4; comp++) {
    for (int j = 0; j < len;) {
      int code = f.get();
      if (code < 0) return false;
      if (code > 128) { // run
        code &= 127;
        int val = f.get();
        while (code-- && j < len) scan[j++][comp] = (unsigned char)val;
      } else { // literal
        while (code-- && j < len) scan[j++][comp] = (unsigned char)f.get();
      }
    }
  }
  return !f.eof;
}

// convertComponent: hdrloader.cpp:120-125 -- (val / 256) * 2^expo, exact
inline float pow2f(int expo) { // (float)pow(2, expo) for expo in [-128, 127], exact
  if (expo >= -126) return ez_u2f((unsigned)(expo + 127) << 23);
  return ez_u2f(1u << (23 + expo + 126)); // 2^-127, 2^-128 are subnormal
}
inline float convert_component(int expo, int val) {
  float v = (float)val / 256.0f;
  float d = pow2f(expo);
  return v * d;
}

} // namespace

bool HDRLoader::loadMemory(const unsigned char* data, size_t len, HDRLoaderResult& res) {
  res.width = res.height = 0;
  res.cols = nullptr;
  if (!data || len < 11 || memcmp(data, "#?RADIANCE", 10) != 0) return false;
  Reader f{data + 11, data + len}; // skip the newline after the magic (hdrloader.cpp:64)
  // header: lines until an empty line
  int c = 0, oldc;
  for (;;) {
    oldc = c;
    c = f.get();
    if (c < 0) return false;
    if (c == 0xa && oldc == 0xa) break;
  }
  char reso[200];
  int i = 0;
  for (;;) {
    c = f.get();
    if (c < 0 || i >= 198) return false;
    reso[i++] = (char)c;
    if (c == 0xa) break;
  }
  reso[i] = 0;
  int w = 0, h = 0;
  if (sscanf(reso, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return false; // %ld-into-int UB fixed
  if ((long long)w * h > (1ll << 28)) return false;
  res.width = w;
  res.height = h;
  float* cols = new float[(size_t)w * h * 3];
  memset(cols, 0, sizeof(float) * (size_t)w * h * 3);
  res.cols = cols;
  std::vector<unsigned char> line((size_t)w * 4);
  RGBE* scan = reinterpret_cast<RGBE*>(line.data());
  for (int y = h - 1; y >= 0; y--) { // scanlines stored in file order: top first
    if (!decrunch(scan, w, f)) break;
    for (int x = 0; x < w; x++) {
      int expo = scan[x][3] - 128;
      cols[0] = convert_component(expo, scan[x][0]);
      cols[1] = convert_component(expo, scan[x][1]);
      cols[2] = convert_component(expo, scan[x][2]);
      cols += 3;
    }
  }
  return true;
}

bool HDRLoader::load(const char* fileName, HDRLoaderResult& res) {
  res.width = res.height = 0;
  res.cols = nullptr;
  FILE* file = fopen(fileName, "rb");
  if (!file) return false;
  std::vector<unsigned char> buf;
  unsigned char chunk[65536];
  size_t n;
  while ((n = fread(chunk, 1, sizeof chunk, file)) > 0) buf.insert(buf.end(), chunk, chunk + n);
  fclose(file);
  return loadMemory(buf.data(), buf.size(), res);
}

// ---------------------------------------------------------------------------
// calculateHdrCache: P5/main.cpp:592-689.  All sums are fp32 running sums in the
// reference's loop order.  Quirks kept: luminance weights
// (0.2, 0.7, 0.1); the table is indexed (row = xi_1*h, col = xi_2*w) although
// the shader looks it up with u = xi_1.  Defined here (UB in the reference):
// lower_bound on the marginal CDF returning `width` is clamped to width-1.

namespace {
// std::lower_bound's halving search, on a strided array
inline int lower_bound_f(const float* a, int n, size_t stride, float val) {
  int first = 0, len = n;
  while (len > 0) {
    int half = len >> 1;
    int mid = first + half;
    if (a[(size_t)mid * stride] < val) {
      first = mid + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  return first;
}
} // namespace

float* calculateHdrCache(const float* HDR, int width, int height) {
  const size_t W = (size_t)width, H = (size_t)height;
  std::vector<float> pdf(W * H);
  float lumSum = 0.0f;
  for (size_t i = 0; i < H; i++)
    for (size_t j = 0; j < W; j++) {
      float R = HDR[3 * (i * W + j)], G = HDR[3 * (i * W + j) + 1], B = HDR[3 * (i * W + j) + 2];
      // `float lum = 0.2 * R + 0.7 * G + 0.1 * B;` -- double literals: evaluated in double, rounded once
      float lum = (float)(0.2 * (double)R + 0.7 * (double)G + 0.1 * (double)B);
      pdf[i * W + j] = lum;
      lumSum += lum;
    }
  for (size_t k = 0; k < W * H; k++) pdf[k] /= lumSum;

  std::vector<float> pdf_x_margin(W, 0.0f);
  for (size_t j = 0; j < W; j++)
    for (size_t i = 0; i < H; i++) pdf_x_margin[j] += pdf[i * W + j];
  std::vector<float> cdf_x_margin = pdf_x_margin;
  for (size_t i = 1; i < W; i++) cdf_x_margin[i] += cdf_x_margin[i - 1];

  // conditional cdf of y given X = j, stored column-major: cdf_y[j*H + i]
  std::vector<float> cdf_y(W * H);
  for (size_t j = 0; j < W; j++) {
    float run = 0.0f;
    for (size_t i = 0; i < H; i++) {
      float cond = pdf[i * W + j] / pdf_x_margin[j];
      run = (i == 0) ? cond : (cond + run); // cdf[i] += cdf[i-1]  ==  cdf[i] = pdf_cond[i] + cdf[i-1]
      cdf_y[j * H + i] = run;
    }
  }

  float* cache = new float[W * H * 3];
  for (size_t j = 0; j < W; j++)
    for (size_t i = 0; i < H; i++) {
      float xi_1 = (float)i / (float)height;
      float xi_2 = (float)j / (float)width;
      int x = lower_bound_f(cdf_x_margin.data(), width, 1, xi_1);
      if (x > width - 1) x = width - 1;
      int y = lower_bound_f(cdf_y.data() + (size_t)x * H, height, 1, xi_2);
      cache[3 * (i * W + j)] = (float)x / (float)width;
      cache[3 * (i * W + j) + 1] = (float)y / (float)height;
      cache[3 * (i * W + j) + 2] = pdf[i * W + j];
    }
  return cache;
}

} // namespace ezrt
