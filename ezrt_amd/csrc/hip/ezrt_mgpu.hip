// ezrt_mgpu.hip -- include/ezrt_mgpu.h: one host process, N devices, one frame.
//
// Layered ON the single-device C ABI (ezrt_scene_create / ezrt_render_device / ...): a replica is an ordinary
// EzrtScene created while its device is current, rendering its shard through EzrtRenderParams.shard_*.  What this
// file adds is the frame close: pack kernel on every peer, ONE grouped ncclSend/ncclRecv to device 0 (or peer / host
// copies), un-permute kernel on device 0.  Design for xGMI (point-to-point, 7 links per GPU): every peer sends its
// shard straight to the root over its own link, once per frame -- 1/N of the frame each, 4 MB per peer for a
// 1024^2 frame on 8 GPUs, ~30 us at link speed; nothing is reduced, so there is nothing for a ring to do.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ezrt_mgpu.h"
#include "ezrt_tiles.h"
#include "ezrt_streams.h"

extern "C" int ezrt_fail_msg(int code, const char* msg); // ezrt_hip.hip: sets ezrt_last_error()

namespace {

int mfail(int code, const std::string& msg) { return ezrt_fail_msg(code, msg.c_str()); }
#define MG_TRY(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) return mfail(EZRT_ERR_DEVICE, std::string(#expr " failed: ") + hipGetErrorString(e_)); \
  } while (0)

// ---- RCCL, bound at run time: the library is only needed by a host that asks for EZRT_TRANSPORT_RCCL, and a
// process that already carries one (PyTorch ships its own librccl.so) must keep using that copy.
typedef struct ncclComm* ncclComm_t;
struct Rccl {
  void* h = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat = 7; // ncclFloat32 (rccl.h)
Rccl g_rccl;
int load_rccl() {
  if (g_rccl.h) return 0;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD); // a copy already in the process
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return mfail(EZRT_ERR_UNSUPPORTED, std::string("EZRT_TRANSPORT_RCCL: cannot load librccl: ") + dlerror());
  Rccl r;
  r.h = h;
#define SYM(field, name)                                                                   \
  *(void**)(&r.field) = dlsym(h, name);                                                    \
  if (!r.field) return mfail(EZRT_ERR_UNSUPPORTED, "EZRT_TRANSPORT_RCCL: librccl lacks " name)
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl = r;
  return 0;
}
#define NCCL_TRY(expr)                                                                                        \
  do {                                                                                                        \
    int e_ = (expr);                                                                                          \
    if (e_ != 0) return mfail(EZRT_ERR_DEVICE, std::string(#expr " failed: ") + g_rccl.GetErrorString(e_));    \
  } while (0)

// ---- kernels: a rank's tiles <-> its packed shard (include/ezrt_tiles.h), one thread per RGBA texel
__global__ void pack_tiles_kernel(const float4* accum, EzrtTilePlan plan, int rank, size_t n, float4* packed) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int x, y;
  const bool in = ezrt_tiles_packed_to_pixel(&plan, rank, k, &x, &y);
  packed[k] = in ? accum[(size_t)y * plan.width + x] : make_float4(0, 0, 0, 0);
}
__global__ void unpack_tiles_kernel(const float4* packed, EzrtTilePlan plan, int rank, size_t n, float4* accum) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int x, y;
  if (ezrt_tiles_packed_to_pixel(&plan, rank, k, &x, &y)) accum[(size_t)y * plan.width + x] = packed[k];
}

// Error paths return from the middle of a multi-device sequence (MG_TRY / NCCL_TRY): these guards put the process back
// into a consistent state on every exit -- the caller's current device restored, an open RCCL group closed (an open
// group would swallow every later collective of the process, torch's included).
struct DeviceRestore {
  int prev = 0;
  DeviceRestore() { (void)hipGetDevice(&prev); }
  ~DeviceRestore() { (void)hipSetDevice(prev); }
};
struct RcclGroup {
  bool open = false;
  int start() {
    const int e = g_rccl.GroupStart();
    open = e == 0;
    return e;
  }
  int end() {
    open = false;
    return g_rccl.GroupEnd();
  }
  ~RcclGroup() {
    if (open) (void)g_rccl.GroupEnd();
  }
};

struct Replica {
  int dev = 0;
  EzrtScene* sc = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e_ready = nullptr;
  float4* accum = nullptr;  // full-size frame buffer; only this shard's tiles are meaningful (all of it on the root after a gather)
  float4* packed = nullptr; // this shard, packed (peers only)
  float4* stage = nullptr;  // pinned host staging (EZRT_TRANSPORT_HOST)
  size_t packed_texels = 0;
  float render_ms = 0.0f;
  bool rendered = false;
};

} // namespace

struct EzrtMgpu {
  std::vector<Replica> r;
  int transport = EZRT_TRANSPORT_PEER;
  int width = 0, height = 0;
  EzrtRenderParams last;
  bool have_last = false;
  float4* recv = nullptr; // on the root: the peers' packed shards back to back
  size_t recv_texels = 0;
  std::vector<ncclComm_t> comms;
  hipEvent_t g0 = nullptr, g1 = nullptr;
  float gather_ms = 0.0f;
  int64_t gather_bytes = 0;
};

namespace {

int release_frames(EzrtMgpu* m) {
  for (Replica& q : m->r) {
    MG_TRY(hipSetDevice(q.dev));
    if (q.accum) MG_TRY(hipFree(q.accum));
    if (q.packed) MG_TRY(hipFree(q.packed));
    if (q.stage) MG_TRY(hipHostFree(q.stage));
    q.accum = q.packed = q.stage = nullptr;
    q.packed_texels = 0;
  }
  if (m->recv) {
    MG_TRY(hipSetDevice(m->r[0].dev));
    MG_TRY(hipFree(m->recv));
    m->recv = nullptr;
  }
  m->recv_texels = 0;
  m->width = m->height = 0;
  return 0;
}

} // namespace

extern "C" {

int ezrt_mgpu_create(const float* tri, int n_tri, const float* nodes, int n_nodes, const int* devices, int n_devices,
                     int transport, EzrtMgpu** out) {
  if (!out) return mfail(EZRT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return mfail(EZRT_ERR_INVALID, "need 1..64 devices");
  if (transport != EZRT_TRANSPORT_RCCL && transport != EZRT_TRANSPORT_PEER && transport != EZRT_TRANSPORT_HOST)
    return mfail(EZRT_ERR_INVALID, "unknown transport");
  int n_visible = 0;
  MG_TRY(hipGetDeviceCount(&n_visible));
  for (int i = 0; i < n_devices; i++) {
    if (devices[i] < 0 || devices[i] >= n_visible)
      return mfail(EZRT_ERR_INVALID, "device ordinal " + std::to_string(devices[i]) + " not visible (" + std::to_string(n_visible) + " devices)");
    if (transport == EZRT_TRANSPORT_RCCL)
      for (int j = 0; j < i; j++)
        if (devices[j] == devices[i]) return mfail(EZRT_ERR_INVALID, "EZRT_TRANSPORT_RCCL needs distinct devices");
  }
  if (transport == EZRT_TRANSPORT_RCCL) {
    int rc = load_rccl();
    if (rc) return rc;
  }
  EzrtMgpu* m = new (std::nothrow) EzrtMgpu();
  if (!m) return mfail(EZRT_ERR_NOMEM, "out of memory");
  m->transport = transport;
  m->r.resize((size_t)n_devices);
  int prev = 0;
  (void)hipGetDevice(&prev);
  auto bail = [&](int rc) {
    ezrt_mgpu_destroy(m); // (keeps ezrt_last_error: destroy never fails)
    (void)hipSetDevice(prev);
    return rc;
  };
  for (int i = 0; i < n_devices; i++) {
    Replica& q = m->r[(size_t)i];
    q.dev = devices[i];
    if (hipSetDevice(q.dev) != hipSuccess) return bail(mfail(EZRT_ERR_DEVICE, "hipSetDevice failed"));
    int rc = ezrt_scene_create(tri, n_tri, nodes, n_nodes, &q.sc); // allocates on the current device
    if (rc) return bail(rc);
    int st_dev = 0; // (= q.dev; the library's streams come from and return to its pool: ezrt_streams.h)
    if (ezh::stream_acquire(false, &q.st, &st_dev) != hipSuccess || hipEventCreate(&q.e0) != hipSuccess ||
        hipEventCreate(&q.e1) != hipSuccess || hipEventCreateWithFlags(&q.e_ready, hipEventDisableTiming) != hipSuccess)
      return bail(mfail(EZRT_ERR_DEVICE, "stream/event creation failed"));
  }
  if (hipSetDevice(m->r[0].dev) != hipSuccess || hipEventCreate(&m->g0) != hipSuccess || hipEventCreate(&m->g1) != hipSuccess)
    return bail(mfail(EZRT_ERR_DEVICE, "event creation failed"));
  if (transport == EZRT_TRANSPORT_PEER)
    for (int i = 1; i < n_devices; i++)
      if (m->r[(size_t)i].dev != m->r[0].dev) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->r[0].dev, m->r[(size_t)i].dev) == hipSuccess && can) {
          hipError_t e = hipDeviceEnablePeerAccess(m->r[(size_t)i].dev, 0); // direct xGMI copies; staged through the host otherwise
          if (e != hipSuccess) (void)hipGetLastError();                    // (already enabled is fine)
        }
      }
  if (transport == EZRT_TRANSPORT_RCCL) { // (one device too: its gather sends the frame to itself, see ezrt_mgpu_gather)
    m->comms.assign((size_t)n_devices, nullptr);
    int e = g_rccl.CommInitAll(m->comms.data(), n_devices, devices);
    if (e != 0) {
      m->comms.clear();
      return bail(mfail(EZRT_ERR_DEVICE, std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(e)));
    }
  }
  (void)hipSetDevice(prev);
  *out = m;
  return 0;
}

void ezrt_mgpu_destroy(EzrtMgpu* m) {
  if (!m) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (ncclComm_t c : m->comms)
    if (c) (void)g_rccl.CommDestroy(c);
  for (Replica& q : m->r) {
    (void)hipSetDevice(q.dev);
    if (q.st) (void)hipStreamSynchronize(q.st);
    if (q.sc) ezrt_scene_destroy(q.sc);
    if (q.accum) (void)hipFree(q.accum);
    if (q.packed) (void)hipFree(q.packed);
    if (q.stage) (void)hipHostFree(q.stage);
    if (q.e0) (void)hipEventDestroy(q.e0);
    if (q.e1) (void)hipEventDestroy(q.e1);
    if (q.e_ready) (void)hipEventDestroy(q.e_ready);
    ezh::stream_park(q.st, false, q.dev);
  }
  if (!m->r.empty()) (void)hipSetDevice(m->r[0].dev);
  if (m->recv) (void)hipFree(m->recv);
  if (m->g0) (void)hipEventDestroy(m->g0);
  if (m->g1) (void)hipEventDestroy(m->g1);
  (void)hipSetDevice(prev);
  delete m;
}

#define FOR_EACH_REPLICA(call)                          \
  if (!m) return mfail(EZRT_ERR_INVALID, "NULL argument"); \
  int prev_ = 0;                                        \
  (void)hipGetDevice(&prev_);                           \
  int rc_ = 0;                                          \
  for (Replica & q : m->r) {                            \
    if (hipSetDevice(q.dev) != hipSuccess) {            \
      rc_ = mfail(EZRT_ERR_DEVICE, "hipSetDevice failed"); \
      break;                                            \
    }                                                   \
    rc_ = (call);                                       \
    if (rc_) break;                                     \
  }                                                     \
  (void)hipSetDevice(prev_);                            \
  return rc_

int ezrt_mgpu_set_env(EzrtMgpu* m, const float* hdr, const float* cache, int w, int h, int filter) {
  FOR_EACH_REPLICA(ezrt_scene_set_env(q.sc, hdr, cache, w, h, filter));
}
int ezrt_mgpu_set_sampler(EzrtMgpu* m, int sobol_dims) { FOR_EACH_REPLICA(ezrt_scene_set_sampler(q.sc, sobol_dims)); }
int ezrt_mgpu_set_option(EzrtMgpu* m, const char* name, int value) { FOR_EACH_REPLICA(ezrt_set_option(q.sc, name, value)); }

int ezrt_mgpu_render(EzrtMgpu* m, const EzrtRenderParams* p) {
  if (!m || !p) return mfail(EZRT_ERR_INVALID, "NULL argument");
  if (p->width <= 0 || p->height <= 0) return mfail(EZRT_ERR_INVALID, "width/height must be positive");
  const int n = (int)m->r.size();
  DeviceRestore restore;
  if (p->width != m->width || p->height != m->height) { // new frame size: empty shards
    int rc = release_frames(m);
    if (rc) return rc;
    const size_t texels = (size_t)p->width * p->height;
    for (Replica& q : m->r) {
      hipError_t e = hipSetDevice(q.dev);
      if (e == hipSuccess) e = hipMalloc((void**)&q.accum, texels * sizeof(float4));
      if (e == hipSuccess) e = hipMemsetAsync(q.accum, 0, texels * sizeof(float4), q.st);
      if (e != hipSuccess) { // all replicas or none: a half-allocated set would pass the size test of the next call
        (void)release_frames(m);
        return mfail(EZRT_ERR_DEVICE, std::string("frame buffers of the replicas: ") + hipGetErrorString(e));
      }
    }
    m->width = p->width;
    m->height = p->height;
  }
  int rc = 0;
  for (int i = 0; i < n && !rc; i++) {
    Replica& q = m->r[(size_t)i];
    MG_TRY(hipSetDevice(q.dev));
    EzrtRenderParams s = *p;
    s.shard_index = i;
    s.shard_count = n;
    MG_TRY(hipEventRecord(q.e0, q.st));
    rc = ezrt_render_device(q.sc, &s, reinterpret_cast<float*>(q.accum), q.st); // enqueues only: the devices run concurrently
    if (rc) break;
    MG_TRY(hipEventRecord(q.e1, q.st));
    q.rendered = true;
  }
  if (rc) return rc;
  m->last = *p;
  m->have_last = true;
  return 0;
}

int ezrt_mgpu_gather(EzrtMgpu* m, float* accum_rgba) {
  if (!m) return mfail(EZRT_ERR_INVALID, "NULL argument");
  if (!m->have_last) return mfail(EZRT_ERR_INVALID, "ezrt_mgpu_gather before any ezrt_mgpu_render");
  const int n = (int)m->r.size();
  DeviceRestore restore;
  const EzrtTilePlan plan = ezrt_tile_plan(m->width, m->height, m->last.tile_w, m->last.tile_h, n);
  // the renders first: their time is theirs, the gather's is measured from here
  for (Replica& q : m->r) {
    MG_TRY(hipSetDevice(q.dev));
    MG_TRY(hipStreamSynchronize(q.st));
    if (q.rendered) {
      MG_TRY(hipEventElapsedTime(&q.render_ms, q.e0, q.e1));
      q.rendered = false;
    }
  }
  Replica& root = m->r[0];
  size_t total = 0;
  for (int i = 1; i < n; i++) total += ezrt_tiles_packed_texels(&plan, i);
  if (total > m->recv_texels) {
    MG_TRY(hipSetDevice(root.dev));
    if (m->recv) MG_TRY(hipFree(m->recv));
    m->recv = nullptr;
    MG_TRY(hipMalloc((void**)&m->recv, total * sizeof(float4)));
    m->recv_texels = total;
  }
  MG_TRY(hipSetDevice(root.dev));
  MG_TRY(hipEventRecord(m->g0, root.st));
  // 1. every peer packs its tiles
  for (int i = 1; i < n; i++) {
    Replica& q = m->r[(size_t)i];
    const size_t cnt = ezrt_tiles_packed_texels(&plan, i);
    MG_TRY(hipSetDevice(q.dev));
    if (cnt > q.packed_texels) {
      if (q.packed) MG_TRY(hipFree(q.packed));
      if (q.stage) MG_TRY(hipHostFree(q.stage));
      q.packed = q.stage = nullptr;
      MG_TRY(hipMalloc((void**)&q.packed, cnt * sizeof(float4)));
      if (m->transport == EZRT_TRANSPORT_HOST) MG_TRY(hipHostMalloc((void**)&q.stage, cnt * sizeof(float4), hipHostMallocPortable));
      q.packed_texels = cnt;
    }
    if (cnt == 0) continue;
    MG_TRY(hipStreamWaitEvent(q.st, m->g0, 0)); // (the gather's clock starts on the root)
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, q.st, q.accum, plan, i, cnt, q.packed);
    if (m->transport == EZRT_TRANSPORT_HOST) MG_TRY(hipMemcpyAsync(q.stage, q.packed, cnt * sizeof(float4), hipMemcpyDeviceToHost, q.st));
    MG_TRY(hipEventRecord(q.e_ready, q.st));
  }
  // 2. the shards travel to the root: ONE grouped exchange
  if (m->transport == EZRT_TRANSPORT_RCCL && n == 1) {
    // One device: nothing has to travel, but a host that asked for RCCL gets RCCL -- the frame is packed, sent to ITSELF through
    // a grouped ncclSend / ncclRecv on the communicator ncclCommInitAll made for the one device, and un-permuted back over the
    // same pixels (the same bits).  This is what a one-GPU box can exercise of the transport: the library is
    // found and bound, the communicator exists, a group with a send and a receive completes on the device's stream.
    const size_t cnt = ezrt_tiles_packed_texels(&plan, 0);
    if (cnt) {
      MG_TRY(hipSetDevice(root.dev));
      if (cnt > root.packed_texels) {
        if (root.packed) MG_TRY(hipFree(root.packed));
        root.packed = nullptr;
        MG_TRY(hipMalloc((void**)&root.packed, cnt * sizeof(float4)));
        root.packed_texels = cnt;
      }
      if (cnt > m->recv_texels) {
        if (m->recv) MG_TRY(hipFree(m->recv));
        m->recv = nullptr;
        MG_TRY(hipMalloc((void**)&m->recv, cnt * sizeof(float4)));
        m->recv_texels = cnt;
      }
      hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, root.st, root.accum, plan, 0, cnt, root.packed);
      {
        RcclGroup group;
        NCCL_TRY(group.start());
        NCCL_TRY(g_rccl.Send(root.packed, cnt * 4, kNcclFloat, 0, m->comms[0], root.st));
        NCCL_TRY(g_rccl.Recv(m->recv, cnt * 4, kNcclFloat, 0, m->comms[0], root.st));
        NCCL_TRY(group.end());
      }
      hipLaunchKernelGGL(unpack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, root.st, m->recv, plan, 0, cnt, root.accum);
      total = cnt; // (reported as the gather's payload)
    }
  } else if (m->transport == EZRT_TRANSPORT_RCCL && n > 1) {
    RcclGroup group; // (closed by its destructor when a send / receive below fails)
    NCCL_TRY(group.start());
    size_t off = 0;
    for (int i = 1; i < n; i++) {
      Replica& q = m->r[(size_t)i];
      const size_t cnt = ezrt_tiles_packed_texels(&plan, i);
      if (cnt) {
        NCCL_TRY(g_rccl.Send(q.packed, cnt * 4, kNcclFloat, 0, m->comms[(size_t)i], q.st));
        NCCL_TRY(g_rccl.Recv(m->recv + off, cnt * 4, kNcclFloat, i, m->comms[0], root.st));
      }
      off += cnt;
    }
    NCCL_TRY(group.end());
  } else {
    MG_TRY(hipSetDevice(root.dev));
    size_t off = 0;
    for (int i = 1; i < n; i++) {
      Replica& q = m->r[(size_t)i];
      const size_t cnt = ezrt_tiles_packed_texels(&plan, i);
      if (cnt) {
        MG_TRY(hipStreamWaitEvent(root.st, q.e_ready, 0));
        if (m->transport == EZRT_TRANSPORT_HOST)
          MG_TRY(hipMemcpyAsync(m->recv + off, q.stage, cnt * sizeof(float4), hipMemcpyHostToDevice, root.st));
        else if (q.dev == root.dev)
          MG_TRY(hipMemcpyAsync(m->recv + off, q.packed, cnt * sizeof(float4), hipMemcpyDeviceToDevice, root.st));
        else
          MG_TRY(hipMemcpyPeerAsync(m->recv + off, root.dev, q.packed, q.dev, cnt * sizeof(float4), root.st));
      }
      off += cnt;
    }
  }
  // 3. un-permute on the root: the peers' tiles into the root's frame buffer (its own tiles are already there)
  MG_TRY(hipSetDevice(root.dev));
  {
    size_t off = 0;
    for (int i = 1; i < n; i++) {
      const size_t cnt = ezrt_tiles_packed_texels(&plan, i);
      if (cnt)
        hipLaunchKernelGGL(unpack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, root.st, m->recv + off, plan, i, cnt, root.accum);
      off += cnt;
    }
  }
  MG_TRY(hipEventRecord(m->g1, root.st));
  MG_TRY(hipGetLastError());
  if (accum_rgba)
    MG_TRY(hipMemcpyAsync(accum_rgba, root.accum, (size_t)m->width * m->height * sizeof(float4), hipMemcpyDeviceToHost, root.st));
  MG_TRY(hipStreamSynchronize(root.st));
  for (int i = 1; i < n; i++) { // (RCCL sends complete with the matching receives; make the peers' streams quiet too)
    MG_TRY(hipSetDevice(m->r[(size_t)i].dev));
    MG_TRY(hipStreamSynchronize(m->r[(size_t)i].st));
  }
  MG_TRY(hipSetDevice(root.dev));
  MG_TRY(hipEventElapsedTime(&m->gather_ms, m->g0, m->g1));
  m->gather_bytes = (int64_t)(total * sizeof(float4));
  return 0;
}

int ezrt_mgpu_frame_device(EzrtMgpu* m, float** frame_dev) {
  if (!m || !frame_dev) return mfail(EZRT_ERR_INVALID, "NULL argument");
  if (!m->r[0].accum) return mfail(EZRT_ERR_INVALID, "no frame yet");
  *frame_dev = reinterpret_cast<float*>(m->r[0].accum);
  return 0;
}

int ezrt_mgpu_counters(EzrtMgpu* m, uint64_t out[EZRT_CTR_COUNT]) {
  if (!m || !out) return mfail(EZRT_ERR_INVALID, "NULL argument");
  DeviceRestore restore;
  for (int k = 0; k < EZRT_CTR_COUNT; k++) out[k] = 0;
  for (Replica& q : m->r) {
    uint64_t c[EZRT_CTR_COUNT];
    MG_TRY(hipSetDevice(q.dev));
    int rc = ezrt_counters(q.sc, c);
    if (rc) return rc;
    for (int k = 0; k < EZRT_CTR_COUNT; k++) out[k] += c[k];
  }
  return 0;
}

int ezrt_mgpu_last_ms(EzrtMgpu* m, float* render_ms, float* gather_ms, int64_t* gather_bytes) {
  if (!m) return mfail(EZRT_ERR_INVALID, "NULL argument");
  if (render_ms)
    for (size_t i = 0; i < m->r.size(); i++) render_ms[i] = m->r[i].render_ms;
  if (gather_ms) *gather_ms = m->gather_ms;
  if (gather_bytes) *gather_bytes = m->gather_bytes;
  return 0;
}

// ---- the kernels alone, for a host that owns the exchange (one process per GPU)
int64_t ezrt_tiles_packed_floats(int width, int height, int tile_w, int tile_h, int rank, int world) {
  if (width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world) return -1;
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  return (int64_t)(ezrt_tiles_packed_texels(&plan, rank) * 4);
}
static int tiles_args_ok(const void* a, const void* b, int width, int height, int rank, int world) {
  if (!a || !b) return mfail(EZRT_ERR_INVALID, "NULL argument");
  if (width <= 0 || height <= 0 || world <= 0 || rank < 0 || rank >= world) return mfail(EZRT_ERR_INVALID, "bad frame / rank / world");
  return 0;
}
int ezrt_tiles_pack_device(const float* accum_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                           float* packed_dev, void* stream) {
  int rc = tiles_args_ok(accum_dev, packed_dev, width, height, rank, world);
  if (rc) return rc;
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  const size_t cnt = ezrt_tiles_packed_texels(&plan, rank);
  if (cnt)
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(accum_dev), plan, rank, cnt, reinterpret_cast<float4*>(packed_dev));
  MG_TRY(hipGetLastError());
  return 0;
}
int ezrt_tiles_unpack_device(const float* packed_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                             float* accum_dev, void* stream) {
  int rc = tiles_args_ok(accum_dev, packed_dev, width, height, rank, world);
  if (rc) return rc;
  const EzrtTilePlan plan = ezrt_tile_plan(width, height, tile_w, tile_h, world);
  const size_t cnt = ezrt_tiles_packed_texels(&plan, rank);
  if (cnt)
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(packed_dev), plan, rank, cnt, reinterpret_cast<float4*>(accum_dev));
  MG_TRY(hipGetLastError());
  return 0;
}

} // extern "C"
