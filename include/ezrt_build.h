/* ezrt_build.h -- GPU scene-build entry points of libezrt_hip.so.
 *
 * ezrt_build_lbvh replaces, for scenes where the host build dominates end-to-end time, the pair
 *   buildBVHwithSAH(triangles, nodes, 0, n-1, leaf_n)        P3/main.cpp:457-588, called at 707-715
 *   + the encode loops                                        P3/main.cpp:720-748
 * with a linear BVH built on the GPU (Morton order, Karras 2012).  Input and output use the
 * reference's encoded layouts: triangles as 36 floats (P3/main.cpp:61-72), nodes as 12 floats
 * (74-78; node 0 = the testNode sentinel, root = node 1, child ids > parent id, leaves of at most
 * leaf_n triangles over a contiguous range of tri_out).  It is an ALTERNATIVE builder: tree shape
 * and node numbering differ from the reference's, so images rendered with it agree with the
 * oracle run on the same arrays, not with images rendered from a buildBVHwithSAH tree.
 *
 * tri_out must hold n_tri * 36 floats, nodes_out nodes_capacity * 12 floats; 2 * n_tri nodes always
 * suffice.  *n_nodes receives the node count (including node 0), *build_ms (optional) the device
 * time of the build without the host<->device copies.  Returns 0 or a negative EZRT_ERR_* code
 * (message in ezrt_last_error()). */
#ifndef EZRT_BUILD_H
#define EZRT_BUILD_H

#ifdef __cplusplus
extern "C" {
#endif

int ezrt_build_lbvh(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out, int nodes_capacity,
                    int* n_nodes, float* build_ms);

/* buildBVHwithSAH itself on the GPU: same arguments as ezrt_build_lbvh, and EXACTLY the arrays the host
 * sequence `nodes = {testNode}; buildBVHwithSAH(triangles, nodes, 0, n-1, leaf_n); encode` produces
 * (ezrt::buildBVHwithSAH of include/ezrt_scene.hpp: same triangle order, same nodes, same ids) -- the
 * parity builder, level by level with radix sorts and segmented scans instead of O(nodes) std::sort calls. */
int ezrt_build_sah(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out, int nodes_capacity,
                   int* n_nodes, float* build_ms);
/* buildBVH (the median-split builder, P3/main.cpp:394-454) on the GPU, same contract: exactly the arrays of
 * `nodes = {testNode}; buildBVH(triangles, nodes, 0, n-1, leaf_n); encode`. */
int ezrt_build_median(const float* tri, int n_tri, int leaf_n, float* tri_out, float* nodes_out, int nodes_capacity,
                      int* n_nodes, float* build_ms);

/* Number of HIP devices this process can see (0 when there is none or the runtime cannot initialise): lets a host
 * decide between ezrt_build_sah and the host buildBVHwithSAH -- both produce the same arrays -- without touching a
 * device.  The Python layer picks the GPU builder for scenes of >= 100 000 triangles when this is > 0. */
int ezrt_build_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
