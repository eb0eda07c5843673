/* abi_layout_check.c -- a plain C11 translation unit that includes the public headers the way a C (cgo, JNI,
 * N-API ...) host would, and pins the layout of the one struct that crosses the boundary by value.  Built and run
 * by `make examples`; tests/test_abi.py runs the binary.  The numbers are the contract: a binding generated from
 * include/ezrt.h on another compiler must agree with them. */
#include <stddef.h>
#include <stdio.h>

#include "ezrt.h"
#include "ezrt_build.h"
#include "ezrt_scene_c.h"

_Static_assert(sizeof(EzrtRenderParams) == 136, "EzrtRenderParams is 34 four-byte fields");
_Static_assert(offsetof(EzrtRenderParams, width) == 0, "width");
_Static_assert(offsetof(EzrtRenderParams, x0) == 8, "x0");
_Static_assert(offsetof(EzrtRenderParams, frame0) == 24, "frame0");
_Static_assert(offsetof(EzrtRenderParams, spp) == 28, "spp");
_Static_assert(offsetof(EzrtRenderParams, max_bounce) == 32, "max_bounce");
_Static_assert(offsetof(EzrtRenderParams, integrator) == 36, "integrator");
_Static_assert(offsetof(EzrtRenderParams, eye) == 40, "eye");
_Static_assert(offsetof(EzrtRenderParams, camera_rotate) == 52, "camera_rotate");
_Static_assert(offsetof(EzrtRenderParams, env_clamp) == 116, "env_clamp");
_Static_assert(offsetof(EzrtRenderParams, tile_w) == 120, "tile_w");
_Static_assert(offsetof(EzrtRenderParams, shard_index) == 128, "shard_index");
_Static_assert(offsetof(EzrtRenderParams, shard_count) == 132, "shard_count");
_Static_assert(EZRT_TRI_FLOATS * sizeof(float) == 144, "P3/main.cpp:61-72: 12 vec3");
_Static_assert(EZRT_NODE_FLOATS * sizeof(float) == 48, "P3/main.cpp:74-78: 4 vec3");
_Static_assert(EZRT_CTR_COUNT == 8, "counter slots");

int main(void) {
  /* the entry points are referenced (not called: no GPU needed) so that a renamed symbol fails the link */
  typedef void (*fn_t)(void);
  fn_t fns[] = {(fn_t)ezrt_scene_create, (fn_t)ezrt_scene_destroy, (fn_t)ezrt_scene_set_env, (fn_t)ezrt_render,
                 (fn_t)ezrt_render_device, (fn_t)ezrt_render_paths, (fn_t)ezrt_query_hits, (fn_t)ezrt_tonemap,
                 (fn_t)ezrt_sobol, (fn_t)ezrt_set_option, (fn_t)ezrt_counters, (fn_t)ezrt_last_error,
                 (fn_t)ezrt_backend, (fn_t)ezrt_host_scene_new, (fn_t)ezrt_host_build_bvh};
  printf("ezrt abi ok: sizeof(EzrtRenderParams)=%zu, %zu entry points linked\n", sizeof(EzrtRenderParams),
         sizeof fns / sizeof fns[0]);
  return 0;
}
