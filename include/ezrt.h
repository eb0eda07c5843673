/* ezrt.h -- the drop-in C ABI for EzRT's per-pixel trace on MI355X.
 *
 * The reference (AKGWSB/EzRT) has no plugin/FFI interface: each chapter is a
 * closed main().  The de-facto boundary between its host code and its trace is
 * the GL resource set created in main() and bound in display():
 *
 *   triangles  float[nTri*36]   12 RGB32F texels / triangle   P5/main.cpp:843-862,878-884
 *   nodes      float[nNodes*12]  4 RGB32F texels / node        P5/main.cpp:864-871,887-893
 *   hdrMap     float[W*H*3]     row 0 = top scanline           P5/main.cpp:896-899
 *   hdrCache   float[W*H*3]     importance-sampling cache      P5/main.cpp:901-906
 *   uniforms   frameCounter,width,height,eye,cameraRotate,hdrResolution
 *                                                              P5/main.cpp:717-720,919-923
 *   lastFrame  RGBA32F WxH running mean, origin bottom-left    P5/main.cpp:926-929, fsh:943-947
 *
 * Every entry point below replaces one of those GL interactions; the cited
 * lines are what a maintainer would delete when binding this library instead
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only; caller owns
 * all host buffers; the library copies at create/set time.  All functions
 * return 0 on success or a negative EZRT_ERR_* code and never exit().
 *
 * The same ABI is implemented twice: libezrt_hip.so (hand-written gfx950 HIP
 * kernels -- the product) and oracle/libezrt_oracle.so (plain-C restatement of
 * the reference arithmetic -- test infrastructure only).
 *
 * Numerical contract: every finite value is reproduced on the bits (triangle
 * ids, distances, radiance, the running mean).  NOT part of the contract: the
 * sign and payload of a NaN.  Chapter 5's estimator can evaluate 0/0 in its MIS
 * weights (P5/fsh:754-757 with both pdfs 0 -- the reference does the same); such a
 * sample poisons its pixel's running mean, and the pixel is NaN in both
 * implementations, but x86 and gfx950 produce different quiet NaNs (0x7fc00000 /
 * 0xffc00000).  Compare frames with NaN == NaN (tests/: `_same_bits_or_both_nan`).
 */
#ifndef EZRT_H
#define EZRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EZRT_OK 0
#define EZRT_ERR_INVALID (-1)   /* bad argument / malformed scene arrays          */
#define EZRT_ERR_DEVICE (-2)    /* HIP runtime error (see ezrt_last_error)        */
#define EZRT_ERR_UNSUPPORTED (-3) /* valid but outside this build's limits        */
#define EZRT_ERR_NOMEM (-4)

/* Record sizes of the reference encoding (P3/fsh:27-28: SIZE_TRIANGLE 12,
 * SIZE_BVHNODE 4 texels of 3 floats). */
#define EZRT_TRI_FLOATS 36
#define EZRT_NODE_FLOATS 12

/* Integrators = the pathTracing variants of the three GPU chapters. */
#define EZRT_INTEGRATOR_P3_DIFFUSE 3  /* P3/fsh:376-413  rand() hemisphere, Lambert, env clamp   */
#define EZRT_INTEGRATOR_P4_DISNEY 4   /* P4/fsh:478-517  rand() hemisphere, anisotropic Disney   */
#define EZRT_INTEGRATOR_P5_SOBOL 50   /* P5/fsh:762-807  Sobol+CP hemisphere, isotropic Disney   */
#define EZRT_INTEGRATOR_P5_MIS 51     /* P5/fsh:810-890  BRDF + env importance sampling, MIS     */
/* Beyond the reference: chapter 5's loop with the anisotropic specular lobe the reference only
 * evaluates in chapter 4 (P4/fsh:440-449; commented out in P5/fsh:472-483) evaluated, importance-sampled and
 * priced: SampleBRDF / BRDF_Pdf with GTR2_aniso (oracle/ezrt_oracle.c sample_gtr2_aniso is the specification). */
#define EZRT_INTEGRATOR_P5_MIS_ANISO 52

#define EZRT_FILTER_NEAREST 0 /* P3/P4 textures */
#define EZRT_FILTER_BILINEAR 1 /* P5/main.cpp:196-197 GL_LINEAR */

typedef struct EzrtScene EzrtScene; /* opaque, device-resident */

/* Replaces the uniform block + FBO of display() (P5/main.cpp:697-748). */
typedef struct EzrtRenderParams {
  int32_t width, height;      /* uniforms width/height = image size                  */
  int32_t x0, y0, x1, y1;     /* half-open pixel rect to render; (0,0,w,h) = all     */
  uint32_t frame0;            /* first frameCounter value (P5/main.cpp:719)          */
  uint32_t spp;               /* number of frames = samples per pixel                */
  int32_t max_bounce;         /* P3: 2, P4: 4, P5: 2 (fsh:436 / 540 / 935)           */
  int32_t integrator;         /* EZRT_INTEGRATOR_*                                   */
  float eye[3];               /* uniform eye                                         */
  float camera_rotate[16];    /* uniform cameraRotate, column-major (GL_FALSE)       */
  float env_clamp;            /* >0: env radiance = min(c, env_clamp) (P3/fsh:154)   */
  int32_t tile_w, tile_h;     /* tile grid used for multi-GPU sharding (0 = 32)      */
  int32_t shard_index;        /* this caller renders tiles with                      */
  int32_t shard_count;        /*   tile_id % shard_count == shard_index (0 or 1=all) */
} EzrtRenderParams;

/* Counter slots of ezrt_counters(): exact integers, the inputs of the
 * algorithmic-bytes roofline figure. */
enum {
  EZRT_CTR_RAYS = 0,      /* hitBVH invocations                                    */
  EZRT_CTR_NODE_POPS = 1, /* P: getBVHNode(top)                P5/fsh:266           */
  EZRT_CTR_INNER_POPS = 2,/* I: two child fetches              P5/fsh:280-287       */
  EZRT_CTR_TRI_TESTS = 3, /* T: getTriangle + hitTriangle      P5/fsh:243-244       */
  EZRT_CTR_MAT_FETCH = 4, /* M: getMaterial on closer hit      P5/fsh:247           */
  EZRT_CTR_SAMPLES = 5,   /* pixel-samples (fragment invocations)                  */
  EZRT_CTR_ENV_MAP = 6,   /* texture2D(hdrMap)                                     */
  EZRT_CTR_ENV_CACHE = 7, /* texture2D(hdrCache)                                   */
  EZRT_CTR_COUNT = 8
};

/* glBufferData+glTexBuffer of the two TBOs (P5/main.cpp:878-893).  tri =
 * nTri*36 floats, nodes = nNodes*12 floats, node 0 dummy, root = node 1. */
int ezrt_scene_create(const float* tri, int n_tri, const float* nodes, int n_nodes, EzrtScene** out);
void ezrt_scene_destroy(EzrtScene* s);

/* glTexImage2D of hdrMap and hdrCache + hdrResolution (P5/main.cpp:896-906).
 * cache may be NULL (integrators 3/4/50 never read it). */
int ezrt_scene_set_env(EzrtScene* s, const float* hdr, const float* cache, int w, int h, int filter);

/* Sobol dimensions of the integrators 50/51/52 (bounce b draws dimensions 2b, 2b+1).  8 (default) = the
 * reference's table, P5/fsh:351-353: four bounces, further bounces wrap (d & 7 -- defined here, the shader reads out
 * of bounds).  16 = eight more dimensions from the tutorial's generator (T5 tutorial.md:267-357) on published
 * Joe-Kuo parameters (include/ezrt_sobol_v16.inc), wrap d & 15: a different estimator from the ninth dimension on,
 * the same one for max_bounce <= 4. */
int ezrt_scene_set_sampler(EzrtScene* s, int sobol_dims);

/* One call = spp iterations of display()'s pass1+pass2 (P5/main.cpp:743-744).
 * accum: host RGBA32F [height][width][4], row 0 = bottom; in = running mean
 * after frame0 samples (ignored when frame0 == 0), out = after frame0+spp. */
int ezrt_render(EzrtScene* s, const EzrtRenderParams* p, float* accum_rgba);

/* Same, but accum is a device pointer (HBM-resident lastFrame) and the work is
 * enqueued on `stream` (a hipStream_t, NULL = default stream) without a host
 * sync.  In the oracle library "device" memory is host memory.
 * ONE STREAM PER SCENE: the queues, counters and block list of a render call are
 * scratch owned by the EzrtScene, so calls on one scene must be ordered -- same
 * stream, or the caller synchronises between streams.  Scenes are independent:
 * concurrent streams (or devices, include/ezrt_mgpu.h) take one scene each.
 * WHAT `stream` ORDERS: every access to accum_rgba_dev (the accumulation of the call's samples) is enqueued on `stream`, in
 * call order, behind whatever the caller queued before; work queued behind the call finds the frame complete, and a host
 * that synchronises `stream` has synchronised the call.  The tracing itself touches only the scene's own scratch and may run
 * on streams of the library's own, overlapping the previous call's tail (option "pipeline_calls") -- which no
 * stream-ordered caller can observe. */
int ezrt_render_device(EzrtScene* s, const EzrtRenderParams* p, float* accum_rgba_dev, void* stream);

/* A device-resident lastFrame for hosts without their own device allocator (the GL texture of P5/main.cpp:926-929:
 * created once, accumulated into by every display() call, read back only to present or save).  frame_create returns a
 * zeroed RGBA32F [height][width][4] buffer on the current device; read / write are the synchronising host copies. */
int ezrt_frame_create(int width, int height, float** frame_dev);
int ezrt_frame_destroy(float* frame_dev);
int ezrt_frame_read(const float* frame_dev, int width, int height, float* rgba_host);
int ezrt_frame_write(float* frame_dev, int width, int height, const float* rgba_host);
/* How many pixels of a device-resident frame have a non-finite R, G or B (a poisoned running mean: see "Numerical contract"
 * above -- the reference's 0/0 in misMixWeight, P5/fsh:754-757, is reproduced, and a caller who wants to know which frames
 * carry such pixels need not copy the frame to the host and scan it).  One small kernel on `stream` + a synchronising
 * 8-byte copy; frame_dev may be any RGBA32F [height][width][4] device buffer (ezrt_render_device's accum_rgba_dev). */
int ezrt_frame_nonfinite(const float* frame_dev, int width, int height, void* stream, int64_t* n_pixels);

/* Parity audit: render exactly one frame (p->frame0, spp ignored) and report
 * for every pixel of the rect and every ray slot of its path the hit triangle
 * id (-1 = miss, -2 = ray not shot) and distance.  Slots: 0 = primary; for
 * bounce b: 1+2b = env shadow ray (integrator 51 only), 2+2b = bounce ray.
 * tri_id/t_hit: host [height][width][1+2*max_bounce].  colour: host RGB
 * [height][width][3] sample radiance (before accumulation), may be NULL. */
int ezrt_render_paths(EzrtScene* s, const EzrtRenderParams* p, int32_t* tri_id, float* t_hit, float* colour);

/* hitBVH on caller-supplied rays (P2/main.cpp:581-588 probe; P5/fsh:254-306).
 * rays: n*6 floats (origin, direction).  tri_id = -1 on miss. */
int ezrt_query_hits(EzrtScene* s, const float* rays_od6, int n_rays, int32_t* tri_id, float* t_hit);

/* pass3: toneMapping(c, 1.5) + pow(1/2.2) (P5/shaders/pass3.fsh:14-24), then
 * P1-style 8-bit quantisation clamp(x*255, 0, 255) (P1/main.cpp:187-189).
 * rgba: host [n_pixels][4] -> rgb8 [n_pixels][3]. */
int ezrt_tonemap(const float* rgba, int n_pixels, uint8_t* rgb8);

/* Sobol generator of the trace (P5/fsh:351-376): out[i*n_dims+d] =
 * sobol(d, grayCode(index0+i)), d < n_dims <= 16 (dimensions 8-15: see ezrt_scene_set_sampler). */
int ezrt_sobol(uint32_t index0, int n, int n_dims, float* out);

/* Schedule knobs of the implementation ("leaf_threshold", "refill_min", "megakernel", ... -- listed in
 * DESIGN.md).  They change how the work is scheduled on the GPU, never the results; unknown names
 * are an error.  The oracle accepts and ignores every name. */
int ezrt_set_option(EzrtScene* s, const char* name, int value);

/* Instrumentation: level 0 counts rays + samples only (timed runs), level 1
 * also counts P/I/T/M and env lookups.  Counters accumulate until reset. */
int ezrt_set_instrumentation(EzrtScene* s, int level);
int ezrt_counters(EzrtScene* s, uint64_t out[EZRT_CTR_COUNT]);
int ezrt_counters_reset(EzrtScene* s);

/* Device time of the kernels of the last ezrt_render* call on this scene, in
 * milliseconds, measured with hipEvents on the launch stream; total and the
 * trace launches alone -- the latter only with ezrt_set_option(s, "launch_events", 1) set before the call (a pair of
 * events around every trace launch; off by default, each record costs the stream a few microseconds), 0 otherwise.
 * total_ms is the span between two events on the CALLER's stream around the call.  With launch_events set the call's chunks run
 * one at a time on that stream (no overlap across chunks or calls), so trace_kernel_ms <= total_ms and both describe the same
 * serial schedule; without it consecutive chunks and calls overlap (pipeline_calls) and total_ms of a call that was queued
 * behind another one includes the wait for it.
 * Forces a sync on the events. */
int ezrt_last_render_ms(EzrtScene* s, float* total_ms, float* trace_kernel_ms, int* n_trace_launches);

/* Scene statistics computed at create time: [0] nTri [1] nNodes [2] tree depth
 * [3] nLeaves [4] max leaf size [5] device bytes. */
int ezrt_scene_stats(EzrtScene* s, int64_t out[6]);

/* What the scene allows the trace's distance pruning to do (a schedule matter: results never depend on it; the proof
 * and its per-triangle bound are in ezrt_amd/csrc/hip/ezrt_traceq4.h): [0] pruning mode in use (0 off, 1 prune, 2 prune +
 * nearest slot first; -1: not available for this scene -- a leaf box that does not hold its triangles, or boxes that are
 * not nested) [1] max 1/sin(theta'/2) over the triangles with a bound [2] max distance of a vertex from its triangle's
 * stored plane [3] max |coordinate| [4] triangles with a large bound or none (slivers: the records above them are never
 * pruned) [5] the launch's margin coefficient a [6] 1 if the device's 4-wide records were built over the library's own SAH tree
 * of the reference's LEAVES (EZRT_RETREE, default; the leaves, and so the results, are the reference's) [7] number of 4-wide
 * records.  The oracle (which never prunes) reports -1 and zeros. */
int ezrt_scene_prune_info(EzrtScene* s, double out[8]);

/* Evaluate the deterministic math definitions on the implementation's compute
 * device (GPU for libezrt_hip) for the bit-equality test.  op: 0 sin, 1 cos,
 * 2 atan2(a,b), 3 asin, 4 log, 5 exp, 6 pow(a,b), 7 sqrt, 8 a/b, 9 wang-hash
 * float of uint bits(a).  Ops 10-12 audit the intersector (a = n rays of 6 floats, b = n boxes of 6 / triangles of
 * 9 floats); ops 13-16 audit integrator 52's sampler in the frame N = (0,0,1) (b = n x (roughness, anisotropic,
 * metallic, clearcoat, clearcoatGloss, -)): 13: a = n x (V, L) -> its pdf; 14/15/16: a = n x (xi1, xi2, xi3, V)
 * -> x / y / z of the sampled direction.  Op 17: out bits = floor(bits(a[i]) / bits(b[0])), 32-bit unsigned, computed
 * the way the kernels divide by launch-invariant counts (exact for every operand).  Op 18 (n >= 2; a is ignored): the
 * implementation's reciprocal 1.0f / x checked against IEEE division for ALL 2^32 bit patterns of x on its compute device
 * -- out[0] = number of mismatches (a NaN equals a NaN), out bits [1] = the first mismatching pattern; the oracle divides, so 0. */
int ezrt_debug_math(int op, const float* a, const float* b, int n, float* out);

const char* ezrt_last_error(void);
/* Optional: release what the library keeps between scenes -- the HIP streams of destroyed scenes, which are parked for
 * the next scene on the same device instead of destroyed.  Returns the number of streams destroyed.  Call
 * it before hipDeviceReset, or to hand the streams back when no further scene will be created. */
int ezrt_trim(void);
const char* ezrt_backend(void); /* "hip:gfx950" or "oracle:cpu" */

#ifdef __cplusplus
}
#endif
#endif /* EZRT_H */
