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

// The two streams the chunks of a render call alternate between (ezrt_render_device, "pipeline_calls") are ONE PAIR PER DEVICE,
// shared by every scene of the process, created together on first use and kept.  Why not a pair per scene (round 4, measured --
// tools/exp_stream_pressure.py, profiles/r4/stream_pressure.txt): the runtime maps streams onto a handful of hardware queues,
// least-loaded first, and a stream created when the other queues are taken shares a queue with the CALLER's stream.  The caller's
// stream carries the accumulation kernels, each behind a wait for its chunk; a barrier waiting in a hardware queue blocks everything
// submitted to that queue after it, so the next chunk -- submitted to the internal stream that shares the queue -- cannot start until
// the previous one is done: the overlap is gone and the event traffic remains (C3: 32.2 -> 31.2 ms per call with no other scene
// alive, 32.2 -> 33.3 with one, -> 31.4 with three, -> 33.4 with three and GPU_MAX_HW_QUEUES=8).  One pair created right after the
// process's first stream gets two queues of its own and keeps them.  Scenes sharing the pair stay independent: a stream orders the
// chunks queued to it, nothing in one scene's chunk waits for another scene's.
// (round 5 made the pair four streams for deeper pipelines -- measured no gain, and unused streams raise the chance that a chunk stream
// shares a hardware queue with the caller's in hosts that use torch or RCCL: a pair again since round 6)
constexpr int SHARED_STREAMS = 2;
struct SharedPair {
  int device;
  hipStream_t st[SHARED_STREAMS];
  int users; // scenes holding the set (ezrt_trim destroys a set nobody holds)
};
inline std::vector<SharedPair> g_shared_pairs; // (guarded by g_stream_pool_mu)
inline hipError_t stream_shared_pair(hipStream_t out[SHARED_STREAMS], int* device) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  *device = dev;
  std::lock_guard<std::mutex> lock(g_stream_pool_mu);
  for (SharedPair& q : g_shared_pairs)
    if (q.device == dev) {
      q.users++;
      for (int i = 0; i < SHARED_STREAMS; i++) out[i] = q.st[i];
      return hipSuccess;
    }
  SharedPair q;
  q.device = dev;
  q.users = 1;
  for (int i = 0; i < SHARED_STREAMS; i++) {
    e = hipStreamCreateWithFlags(&q.st[i], hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (int j = 0; j < i; j++) (void)hipStreamDestroy(q.st[j]);
      return e;
    }
  }
  g_shared_pairs.push_back(q);
  for (int i = 0; i < SHARED_STREAMS; i++) out[i] = q.st[i];
  return hipSuccess;
}

inline void stream_shared_release(int device) {
  std::lock_guard<std::mutex> lock(g_stream_pool_mu);
  for (SharedPair& q : g_shared_pairs)
    if (q.device == device && q.users > 0) q.users--;
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
    for (size_t i = 0; i < g_shared_pairs.size();) { // the shared pairs no scene holds any more
      if (g_shared_pairs[i].users > 0) {
        i++;
        continue;
      }
      for (int k = 0; k < SHARED_STREAMS; k++) all.push_back({g_shared_pairs[i].device, false, g_shared_pairs[i].st[k]});
      g_shared_pairs.erase(g_shared_pairs.begin() + (long)i);
    }
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
