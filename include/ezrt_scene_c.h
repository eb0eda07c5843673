/* ezrt_scene_c.h -- C shim over the C++ host scene-build API of
 * include/ezrt_scene.hpp (libezrt_scene.so), so Python (ctypes), C and other
 * FFI hosts can drive the reference's scene-build sequence
 * (readObj -> buildBVHwithSAH -> encode, P3/main.cpp:676-748) without a C++
 * compiler.  Pure host code: no HIP dependency.  Functions return 0 or a
 * negative code; ezrt_host_last_error() holds the message. */
#ifndef EZRT_SCENE_C_H
#define EZRT_SCENE_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct EzrtHostScene EzrtHostScene; /* std::vector<Triangle> + std::vector<BVHNode> */

EzrtHostScene* ezrt_host_scene_new(void);
void ezrt_host_scene_free(EzrtHostScene* h);

/* Material as 18 floats in struct order (emissive3, baseColor3, subsurface,
 * metallic, specular, specularTint, roughness, anisotropic, sheen, sheenTint,
 * clearcoat, clearcoatGloss, IOR, transmission).  which: 3 = P3 defaults,
 * 4 = P4/P5 defaults. */
int ezrt_host_material_defaults(int which, float out18[18]);

/* getTransformMatrix (P3/main.cpp:254-270); out = column-major mat4 */
int ezrt_host_get_transform_matrix(const float rot_deg[3], const float translate[3], const float scale[3],
                                   float out16[16]);

/* readObj (P3/main.cpp:273-391): appends triangles */
int ezrt_host_read_obj(EzrtHostScene* h, const char* path, const float material18[18], const float trans16[16],
                       int smooth_normal);
int ezrt_host_read_obj_text(EzrtHostScene* h, const char* text, int64_t len, const float material18[18],
                            const float trans16[16], int smooth_normal);
/* append already-built triangles given in the encoded 36-float layout */
int ezrt_host_add_triangles(EzrtHostScene* h, const float* tri36, int n);

/* nodes = {testNode}; buildBVH / buildBVHwithSAH(triangles, nodes, 0, n-1, leaf_n)
 * (P3/main.cpp:707-715).  method: 0 = median (buildBVH), 1 = SAH. */
int ezrt_host_build_bvh(EzrtHostScene* h, int method, int leaf_n);
/* order of triangles with exactly equal sort keys (per calling thread): 0 = stable (default),
 * 1 = this toolchain's std::sort = what the reference's std::sort does when built here */
int ezrt_host_set_tie_order(int library_sort);
/* [0] inf-cap fallback nodes [1] std::sort calls [2] max depth (root = 1) */
int ezrt_host_build_stats(EzrtHostScene* h, int64_t out[3]);

int ezrt_host_counts(EzrtHostScene* h, int* n_tri, int* n_nodes);
/* encode loops (P3/main.cpp:720-748): tri_out[nTri*36], nodes_out[nNodes*12] */
int ezrt_host_encode(EzrtHostScene* h, float* tri_out, float* nodes_out);

/* HDRLoader::load (P5/lib/hdrloader.cpp:50-118).  *data is malloc-ed w*h*3
 * floats, released with ezrt_host_free. */
int ezrt_host_hdr_load(const char* path, int* w, int* h, float** data);
int ezrt_host_hdr_load_memory(const unsigned char* bytes, int64_t len, int* w, int* h, float** data);
/* calculateHdrCache (P5/main.cpp:592-689) into caller memory out[w*h*3] */
int ezrt_host_hdr_cache(const float* hdr, int w, int h, float* out);
void ezrt_host_free(void* p);

/* eye + cameraRotate from (rotatAngle, upAngle, r) (P3/main.cpp:607-610) */
int ezrt_host_camera(float rotat_angle_deg, float up_angle_deg, float r, float eye3[3], float camera_rotate16[16]);

/* Chapter 2's CPU query (ezrt::p2 in ezrt_scene.hpp; P2/main.cpp:242-485) in one call:
 * triangles as 9 floats (p1,p2,p3) -> buildBVH (method 0) / buildBVHwithSAH (1) with leaf size
 * leaf_n -> per ray (origin3, direction3) hitBVH (use_bvh 1) or hitTriangleArray over everything (0).
 * hit_index = index into the builder-sorted triangle array (returned in tri_sorted9 when not NULL)
 * or -1; hit_t = distance or INF = 114514. */
int ezrt_host_p2_query(const float* tri9, int n_tri, int method, int leaf_n, const float* rays6, int n_rays,
                       int use_bvh, float* tri_sorted9, int* hit_index, float* hit_t);

const char* ezrt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
