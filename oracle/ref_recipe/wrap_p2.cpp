/* TEST INFRASTRUCTURE -- oracle/_ref: chapter 2 of the REFERENCE, compiled from where it lies.
 *
 * #includes `part 2 -- BVH Accelerate Struct/source code/main.cpp` unmodified (EZRT_REF_MAIN,
 * passed by build_ref.py) against the headless GL/GLM stand-ins in shim/ and exports the
 * reference's C++ twins of the intersector -- the only CPU code in the reference that computes
 * what the shader's hitTriangle / hitAABB / hitArray compute:
 *   hitTriangle P2/main.cpp:212-238   hitAABB 449-463   hitTriangleArray 436-446
 *   hitBVH 466-485 (NOTE its leaf range `root->n, root->n + root->index - 1` swaps n and index:
 *   exported as it is, a test documents the bug; the shader's range is P3/fsh:334-337)
 *   buildBVH 242-294 / buildBVHwithSAH 297-433 (pointer tree), main()'s probe ray 581-588.
 */
#include <unistd.h>
#include <cstring>
#include <vector>

#define main ezrt_ref_chapter_main
#include EZRT_REF_MAIN
#undef main

static Ray ray_from(const float* r6) {
    Ray r;
    r.startPoint = vec3(r6[0], r6[1], r6[2]);
    r.direction = vec3(r6[3], r6[4], r6[5]);
    return r;
}
static BVHNode* g_root = nullptr;

static void flatten(BVHNode* n, std::vector<int>& ints, std::vector<float>& boxes) {
    if (!n) return;
    size_t me = ints.size() / 4;
    ints.insert(ints.end(), {0, 0, n->left ? 0 : n->n, n->left ? 0 : n->index});
    boxes.insert(boxes.end(), {n->AA.x, n->AA.y, n->AA.z, n->BB.x, n->BB.y, n->BB.z});
    if (n->left) { ints[4 * me + 0] = int(ints.size() / 4); flatten(n->left, ints, boxes); }
    if (n->right) { ints[4 * me + 1] = int(ints.size() / 4); flatten(n->right, ints, boxes); }
}

extern "C" {

float ref_p2_inf() { return float(INF); }

void ref_p2_set_triangles(const float* tri9, int n) {
    triangles.clear();
    for (int i = 0; i < n; i++) {
        const float* t = tri9 + 9 * i;
        triangles.push_back(Triangle(vec3(t[0], t[1], t[2]), vec3(t[3], t[4], t[5]), vec3(t[6], t[7], t[8])));
    }
    g_root = nullptr;
}
int ref_p2_count() { return int(triangles.size()); }
void ref_p2_get_triangles(float* tri9) {
    for (size_t i = 0; i < triangles.size(); i++) {
        std::memcpy(tri9 + 9 * i, &triangles[i].p1, 12);
        std::memcpy(tri9 + 9 * i + 3, &triangles[i].p2, 12);
        std::memcpy(tri9 + 9 * i + 6, &triangles[i].p3, 12);
    }
}

/* pointer tree over the global array (sorted in place); returns the node count, pre-order */
int ref_p2_build(int sah, int leaf_n) {
    int last = int(triangles.size()) - 1;
    g_root = sah ? buildBVHwithSAH(triangles, 0, last, leaf_n) : buildBVH(triangles, 0, last, leaf_n);
    std::vector<int> ints; std::vector<float> boxes;
    flatten(g_root, ints, boxes);
    return int(ints.size() / 4);
}
void ref_p2_get_tree(int* ints4, float* boxes6) {
    std::vector<int> ints; std::vector<float> boxes;
    flatten(g_root, ints, boxes);
    std::memcpy(ints4, ints.data(), ints.size() * sizeof(int));
    std::memcpy(boxes6, boxes.data(), boxes.size() * sizeof(float));
}

void ref_p2_hit_triangle(const float* tri9, int n, const float* rays6, float* t) {
    /* ray i against triangle i */
    for (int i = 0; i < n; i++) {
        const float* p = tri9 + 9 * i;
        Triangle tr(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]), vec3(p[6], p[7], p[8]));
        t[i] = hitTriangle(&tr, ray_from(rays6 + 6 * i));
    }
}
void ref_p2_hit_aabb(const float* rays6, const float* aabb6, int n, float* t) {
    for (int i = 0; i < n; i++) {
        const float* b = aabb6 + 6 * i;
        t[i] = hitAABB(ray_from(rays6 + 6 * i), vec3(b[0], b[1], b[2]), vec3(b[3], b[4], b[5]));
    }
}
/* brute force over the whole global array: index of the winner (or -1) and its distance */
void ref_p2_hit_triangle_array(const float* rays6, int m, int* idx, float* t) {
    #pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < m; i++) {
        HitResult r = hitTriangleArray(ray_from(rays6 + 6 * i), triangles, 0, int(triangles.size()) - 1);
        idx[i] = r.triangle ? int(r.triangle - &triangles[0]) : -1;
        t[i] = r.distance;
    }
}
void ref_p2_hit_bvh(const float* rays6, int m, int* idx, float* t) {
    for (int i = 0; i < m; i++) {
        HitResult r = hitBVH(ray_from(rays6 + 6 * i), triangles, g_root);
        idx[i] = r.triangle ? int(r.triangle - &triangles[0]) : -1;
        t[i] = r.distance;
    }
}

/* chapter 2's main() headless (cwd = its source directory).  Afterwards the global `triangles`
 * is the SAH-sorted scene and the last 14 entries of `lines` are the probe ray's winner
 * (addTriangle, 12 points) and the ray itself (addLine). */
int ref_p2_run_main(const char* source_dir) {
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return -1;
    if (chdir(source_dir) != 0) return -2;
    triangles.clear(); vertices.clear(); indices.clear(); lines.clear();
    std::cout.setstate(std::ios_base::failbit);
    char arg0[] = "ezrt_ref";
    char* argv[] = {arg0, nullptr};
    int rc = ezrt_ref_chapter_main(1, argv);
    std::cout.clear();
    if (chdir(cwd) != 0) return -3;
    return rc;
}
int ref_p2_lines_count() { return int(lines.size()); }
void ref_p2_lines_get(float* out3) { std::memcpy(out3, lines.data(), lines.size() * 12); }

}  // extern "C"
