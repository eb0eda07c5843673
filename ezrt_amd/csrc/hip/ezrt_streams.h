// Process-wide pool of the library's own HIP streams (host code; shared by ezrt_hip.hip and ezrt_mgpu.hip).
//
// The library's streams are PARKED when their owner (a scene, a multi-device context) is destroyed and handed to the
// next owner on the same device, never destroyed -- as torch does with its stream pool.  Measured on MI355X / ROCm 7.2:
// after hipStreamDestroy of a scene's streams, streams created afterwards in the same process make the cross-stream
// event waits of a frame (two per stage: main launch -> redo launch -> second shading pass) slow -- a second scene
// rendered 15-35 % slower than the first (C4 3.42 -> 4.12 ms per call, C2 2.37 -> 3.24) with the same kernels and
// buffers; destroying only the events, or only freeing the buffers, costs nothing.  With the streams kept, scene after
// scene runs at the first one's speed (tools/exp_seq.py).  The pool grows to the largest number of streams in use at once.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

namespace ezh {

struct PooledStream {
  int device;
  bool high_priority;
  hipStream_t st;
};
inline std::mutex g_stream_pool_mu;
inline std::vector<PooledStream> g_stream_pool;

// A non-blocking stream on the CURRENT device (highest priority if asked); *device = that device, for stream_park.
inline hipError_t stream_acquire(bool high_priority, hipStream_t* out, int* device) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  *device = dev;
  {
    std::lock_guard<std::mutex> lock(g_stream_pool_mu);
    for (size_t i = 0; i < g_stream_pool.size(); i++) {
      if (g_stream_pool[i].device != dev || g_stream_pool[i].high_priority != high_priority) continue;
      *out = g_stream_pool[i].st;
      g_stream_pool.erase(g_stream_pool.begin() + (long)i);
      return hipSuccess;
    }
  }
  if (!high_priority) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio_hi);
}

// Work still queued on the stream simply finishes; the next owner's work queues behind it.
inline void stream_park(hipStream_t st, bool high_priority, int device) {
  if (!st) return;
  std::lock_guard<std::mutex> lock(g_stream_pool_mu);
  g_stream_pool.push_back({device, high_priority, st});
}

// ezrt_trim: destroy every parked stream (the application is about to reset the device, or is done creating scenes).
inline int stream_pool_trim() {
  std::vector<PooledStream> all;
  {
    std::lock_guard<std::mutex> lock(g_stream_pool_mu);
    all.swap(g_stream_pool);
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (const PooledStream& q : all) {
    (void)hipSetDevice(q.device);
    (void)hipStreamDestroy(q.st);
  }
  (void)hipSetDevice(prev);
  return (int)all.size();
}

inline size_t stream_pool_size() {
  std::lock_guard<std::mutex> lock(g_stream_pool_mu);
  return g_stream_pool.size();
}

} // namespace ezh
