/* ezrt_mgpu.h -- closing a frame over several GPUs from ONE host process: the multi-device form of
 * include/ezrt.h for a C/C++ host such as the reference's own display() loop (P5/main.cpp:697-748), which has one
 * process, one scene and one lastFrame.  (A one-process-per-GPU host -- bench.py under torch.distributed -- uses
 * EzrtRenderParams.shard_* with ezrt_render_device and the two ezrt_tiles_* kernels below around its own gather.)
 *
 *   reference                                      here
 *   one GL context, one lastFrame target           N device replicas of the scene, N lastFrame shards: device i
 *   (P5/main.cpp:926-929)                          accumulates the tiles t with t % N == i (include/ezrt_tiles.h)
 *   pass1.draw(); pass2.draw() per frame           ezrt_mgpu_render: every device traces all spp of its tiles,
 *   (P5/main.cpp:743-744)                          concurrently, each on its own stream; no data-path collective
 *   glutSwapBuffers / reading lastFrame            ezrt_mgpu_gather: each peer packs its tiles, ONE grouped
 *                                                  ncclSend/ncclRecv (RCCL over xGMI) brings them to device 0,
 *                                                  an un-permute kernel there writes them into the full frame
 *
 * Pixels are independent and every device runs the same per-pixel arithmetic, so the assembled frame is
 * bit-identical to the one-GPU frame of ezrt_render (tests/test_gpu_mgpu.py).  Errors: 0 or a negative EZRT_ERR_*
 * code, message in ezrt_last_error().  One host thread per EzrtMgpu. */
#ifndef EZRT_MGPU_H
#define EZRT_MGPU_H

#include "ezrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct EzrtMgpu EzrtMgpu;

#define EZRT_TRANSPORT_RCCL 0 /* grouped ncclSend / ncclRecv to device 0 (librccl, loaded on first use); distinct devices */
#define EZRT_TRANSPORT_PEER 1 /* hipMemcpyPeerAsync device i -> device 0                                                  */
#define EZRT_TRANSPORT_HOST 2 /* staged through pinned host memory (no peer access needed; the test transport)           */

/* The scene arrays of ezrt_scene_create replicated on `n_devices` HIP devices (ordinals in `devices`; device[0] is the
 * root that assembles the frame).  An ordinal may repeat -- several shards on one GPU, for tests and for a node
 * with fewer GPUs than shards -- except with EZRT_TRANSPORT_RCCL. */
int ezrt_mgpu_create(const float* tri, int n_tri, const float* nodes, int n_nodes, const int* devices, int n_devices,
                     int transport, EzrtMgpu** out);
void ezrt_mgpu_destroy(EzrtMgpu* m);
/* ezrt_scene_set_env / ezrt_scene_set_sampler / ezrt_set_option on every replica */
int ezrt_mgpu_set_env(EzrtMgpu* m, const float* hdr, const float* cache, int w, int h, int filter);
int ezrt_mgpu_set_sampler(EzrtMgpu* m, int sobol_dims);
int ezrt_mgpu_set_option(EzrtMgpu* m, const char* name, int value);

/* spp more frames on every device (asynchronous: returns once the work is enqueued).  p->shard_index/shard_count are
 * ignored (device i renders shard i of n_devices); p->tile_w/tile_h choose the tile size; frame0 > 0 continues the
 * running mean of the device-resident shards.  A change of width/height starts empty shards. */
int ezrt_mgpu_render(EzrtMgpu* m, const EzrtRenderParams* p);

/* Close the frame: wait for the renders, pack -> transport -> un-permute on device 0.  accum_rgba (host, [h][w][4],
 * row 0 = bottom) receives the assembled running mean when not NULL.  The shards stay valid: further
 * ezrt_mgpu_render calls continue them. */
int ezrt_mgpu_gather(EzrtMgpu* m, float* accum_rgba);
/* device pointer (on devices[0]) of the assembled frame after ezrt_mgpu_gather */
int ezrt_mgpu_frame_device(EzrtMgpu* m, float** frame_dev);

/* counters of ezrt_counters summed over the devices */
int ezrt_mgpu_counters(EzrtMgpu* m, uint64_t out[EZRT_CTR_COUNT]);
/* of the last render + gather: device time of each device's render calls [n_devices], of the gather (first pack
 * launch to the end of the un-permute kernel, after all renders had finished), and the payload that crossed */
int ezrt_mgpu_last_ms(EzrtMgpu* m, float* render_ms, float* gather_ms, int64_t* gather_bytes);

/* ---- the two kernels around a host-owned gather (one process per GPU).  accum_dev: the rank's full-size RGBA32F
 * frame buffer; packed_dev: ezrt_tiles_packed_floats(...) floats (layout: include/ezrt_tiles.h).  Enqueued on
 * `stream` (hipStream_t), no sync. */
int64_t ezrt_tiles_packed_floats(int width, int height, int tile_w, int tile_h, int rank, int world);
int ezrt_tiles_pack_device(const float* accum_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                           float* packed_dev, void* stream);
int ezrt_tiles_unpack_device(const float* packed_dev, int width, int height, int tile_w, int tile_h, int rank, int world,
                             float* accum_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EZRT_MGPU_H */
