/* ezrt_tiles.h -- the index rules of the image-space sharding, shared by the kernels of
 * libezrt_hip.so, the CPU oracle and any host that wants to address a packed shard itself.
 *
 * A W x H frame is cut into tile_w x tile_h tiles numbered row-major from the bottom-left (tile =
 * (y / tile_h) * tiles_x + x / tile_w, the rule of EzrtRenderParams.shard_*, include/ezrt.h); tile t belongs to
 * rank t % world.  A rank's PACKED shard holds its tiles in ascending tile id, each as tile_h rows of tile_w RGBA32F
 * texels (edge tiles are stored whole; texels outside the frame are zero).  This is what crosses xGMI when a frame is
 * closed: the reference has no counterpart (one GPU, one GL target: P5/main.cpp:926-929). */
#ifndef EZRT_TILES_H
#define EZRT_TILES_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define EZRT_TILES_FN static inline __host__ __device__
#else
#define EZRT_TILES_FN static inline
#endif

typedef struct EzrtTilePlan {
  int32_t width, height, tile_w, tile_h; /* tile_w / tile_h: as passed, 0 already replaced by 32 */
  int32_t tiles_x, tiles_y, n_tiles, world;
} EzrtTilePlan;

EZRT_TILES_FN EzrtTilePlan ezrt_tile_plan(int width, int height, int tile_w, int tile_h, int world) {
  EzrtTilePlan p;
  p.width = width;
  p.height = height;
  p.tile_w = tile_w > 0 ? tile_w : 32;
  p.tile_h = tile_h > 0 ? tile_h : 32;
  p.tiles_x = (width + p.tile_w - 1) / p.tile_w;
  p.tiles_y = (height + p.tile_h - 1) / p.tile_h;
  p.n_tiles = p.tiles_x * p.tiles_y;
  p.world = world > 0 ? world : 1;
  return p;
}
/* tiles owned by `rank`: ids rank, rank + world, ... */
EZRT_TILES_FN int ezrt_tiles_owned(const EzrtTilePlan* p, int rank) {
  return rank < p->n_tiles ? (p->n_tiles - rank + p->world - 1) / p->world : 0;
}
/* RGBA32F texels of a packed shard */
EZRT_TILES_FN size_t ezrt_tiles_packed_texels(const EzrtTilePlan* p, int rank) {
  return (size_t)ezrt_tiles_owned(p, rank) * (size_t)p->tile_w * (size_t)p->tile_h;
}
/* packed texel index k of `rank`'s shard -> pixel (x, y); returns 0 when the texel lies outside the frame */
EZRT_TILES_FN int ezrt_tiles_packed_to_pixel(const EzrtTilePlan* p, int rank, size_t k, int* x, int* y) {
  const size_t per_tile = (size_t)p->tile_w * (size_t)p->tile_h;
  const int j = (int)(k / per_tile), r = (int)(k % per_tile);
  const int tile = rank + j * p->world;
  *x = (tile % p->tiles_x) * p->tile_w + r % p->tile_w;
  *y = (tile / p->tiles_x) * p->tile_h + r / p->tile_w;
  return *x < p->width && *y < p->height;
}

#endif /* EZRT_TILES_H */
