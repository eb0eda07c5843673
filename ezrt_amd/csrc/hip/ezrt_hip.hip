// ezrt_hip.hip -- libezrt_hip.so: the C ABI of include/ezrt.h implemented on hand-written gfx950 kernels.  This file: error plumbing,
// scene lifetime, options, counters, device-resident frames.  ezrt_scene_create / ezrt_scene_set_env: ezrt_scene_build.hip; the render,
// audit and utility entry points (everything that launches a kernel): ezrt_launch.hip.  There is no CPU compute path.
#include "ezrt_internal.h"

namespace ezi {
namespace {
thread_local char g_err[512];
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error() { return g_err; }
} // namespace ezi

extern "C" {

const char* ezrt_last_error(void) { return ezi::last_error(); }
// for the other translation units of this library (ezrt_lbvh.hip)
__attribute__((visibility("hidden"))) int ezrt_fail_msg(int code, const char* msg) { return fail(code, "%s", msg); }
const char* ezrt_backend(void) { return "hip:gfx950"; }

int ezrt_trim(void) { return ezh::stream_pool_trim(); }

void ezrt_scene_destroy(EzrtScene* s) {
  if (!s) return;
  // (whatever ensure_events got to create: it may have stopped half way)
  if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
  if (s->ev_end) (void)hipEventDestroy(s->ev_end);
  for (Pipe& q : s->pipe) {
    // (q.stream is the device's shared pair: released below, not parked)
    if (q.ev_done) (void)hipEventDestroy(q.ev_done);
    if (q.ev_free) (void)hipEventDestroy(q.ev_free);
  }
  for (int i = 0; i < MAX_TRACE_EVENTS; i++) {
    if (s->ev_trace[i][0]) (void)hipEventDestroy(s->ev_trace[i][0]);
    if (s->ev_trace[i][1]) (void)hipEventDestroy(s->ev_trace[i][1]);
  }
  if (s->pipe[0].stream) ezh::stream_shared_release(s->pipe[0].stream_device);
  delete s;
}

int ezrt_frame_create(int width, int height, float** frame_dev) {
  if (!frame_dev || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  *frame_dev = nullptr;
  const size_t bytes = (size_t)width * height * sizeof(float4);
  float* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, bytes));
  hipError_t e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(EZRT_ERR_DEVICE, "hipMemset failed: %s", hipGetErrorString(e));
  }
  *frame_dev = p;
  return 0;
}
int ezrt_frame_destroy(float* frame_dev) {
  if (frame_dev) HIP_TRY(hipFree(frame_dev));
  return 0;
}
// Synchronise the device that OWNS the frame (whatever stream rendered into it), not whichever device happens to be
// current: the frame may have been created and rendered while another device was current (ezrt_mgpu, a host that
// switches devices).  The caller's current device is restored on every exit.
static int frame_copy(float* frame_dev, float* rgba_host, size_t bytes, bool to_host) {
  hipPointerAttribute_t at;
  HIP_TRY(hipPointerGetAttributes(&at, frame_dev));
  int prev = 0;
  HIP_TRY(hipGetDevice(&prev));
  struct Restore {
    int d;
    ~Restore() { (void)hipSetDevice(d); }
  } restore{prev};
  if (at.device != prev) HIP_TRY(hipSetDevice(at.device));
  HIP_TRY(hipDeviceSynchronize());
  if (to_host) HIP_TRY(hipMemcpy(rgba_host, frame_dev, bytes, hipMemcpyDeviceToHost));
  else HIP_TRY(hipMemcpy(frame_dev, rgba_host, bytes, hipMemcpyHostToDevice));
  return 0;
}
int ezrt_frame_read(const float* frame_dev, int width, int height, float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  return frame_copy(const_cast<float*>(frame_dev), rgba_host, (size_t)width * height * sizeof(float4), true);
}
int ezrt_frame_write(float* frame_dev, int width, int height, const float* rgba_host) {
  if (!frame_dev || !rgba_host || width <= 0 || height <= 0) return fail(EZRT_ERR_INVALID, "bad frame arguments");
  return frame_copy(frame_dev, const_cast<float*>(rgba_host), (size_t)width * height * sizeof(float4), false);
}

int ezrt_scene_set_sampler(EzrtScene* s, int sobol_dims) {
  if (!s || (sobol_dims != 8 && sobol_dims != 16)) return fail(EZRT_ERR_INVALID, "sobol_dims must be 8 or 16");
  s->sobol_mask = (uint32_t)sobol_dims - 1u;
  return 0;
}

int ezrt_set_option(EzrtScene* s, const char* name, int value) {
  if (!s || !name) return fail(EZRT_ERR_INVALID, "NULL argument");
  for (const TuningName& k : kTuning)
    if (strcmp(k.name, name) == 0) {
      if (value < k.lo || value > k.hi)
        return fail(EZRT_ERR_INVALID, "option '%s' = %d outside [%d, %d]", name, value, k.lo, k.hi);
      s->tune.*(k.field) = value;
      return 0;
    }
  return fail(EZRT_ERR_INVALID, "unknown option '%s'", name);
}

int ezrt_set_instrumentation(EzrtScene* s, int level) {
  if (!s || level < 0 || level > 1) return fail(EZRT_ERR_INVALID, "bad instrumentation level");
  s->instr = level;
  return 0;
}
int ezrt_counters(EzrtScene* s, uint64_t out[EZRT_CTR_COUNT]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long all[CTR_SLOTS * EZRT_CTR_COUNT];
  HIP_TRY(hipMemcpy(all, s->counters.p, sizeof all, hipMemcpyDeviceToHost));
  for (int k = 0; k < EZRT_CTR_COUNT; k++) {
    out[k] = 0;
    for (int j = 0; j < CTR_SLOTS; j++) out[k] += all[j * EZRT_CTR_COUNT + k];
  }
  return 0;
}
int ezrt_counters_reset(EzrtScene* s) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(s->counters.p, 0, (size_t)CTR_SLOTS * EZRT_CTR_COUNT * sizeof(unsigned long long)));
  return 0;
}
int ezrt_last_render_ms(EzrtScene* s, float* total_ms, float* trace_kernel_ms, int* n_trace_launches) {
  if (!s) return fail(EZRT_ERR_INVALID, "NULL argument");
  if (!s->timed) return fail(EZRT_ERR_INVALID, "no render call to time yet");
  HIP_TRY(hipEventSynchronize(s->ev_end));
  float tot = 0.0f, tr = 0.0f;
  HIP_TRY(hipEventElapsedTime(&tot, s->ev_begin, s->ev_end));
  for (int i = 0; i < s->n_trace_events; i++) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_trace[i][0], s->ev_trace[i][1]));
    tr += ms;
  }
  if (total_ms) *total_ms = tot;
  if (trace_kernel_ms) *trace_kernel_ms = tr;
  if (n_trace_launches) *n_trace_launches = s->n_trace_launches;
  return 0;
}
int ezrt_scene_prune_info(EzrtScene* s, double out[8]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  out[0] = s->prunable ? (double)prune_mode(s) : -1.0;
  out[1] = s->prune_G;
  out[2] = s->prune_Z;
  out[3] = s->prune_M;
  out[4] = (double)s->prune_bad;
  out[5] = (double)s->prune_a;
  out[6] = s->retreed ? 1.0 : 0.0;
  out[7] = (double)s->n_inner4;
  return 0;
}
int ezrt_scene_stats(EzrtScene* s, int64_t out[6]) {
  if (!s || !out) return fail(EZRT_ERR_INVALID, "NULL argument");
  memcpy(out, s->stats, sizeof s->stats);
  return 0;
}

} // extern "C"
