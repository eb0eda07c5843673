/* TEST INFRASTRUCTURE -- oracle/_ref: the REFERENCE ITSELF, compiled from where it lies.
 *
 * This translation unit #includes one of the reference's chapter programs unmodified
 *   EZRT_REF_MAIN    = ".../part {3,4,5} .../source code/main.cpp"
 * (and links wrap_hdr.cpp = the chapter's lib/hdrloader.cpp)
 * (paths are passed by oracle/ref_recipe/build_ref.py; nothing is copied into this repository)
 * against the headless GL/GLM stand-ins in shim/, renames its main(), and exports the reference's
 * own host functions through a small C interface for the parity tests:
 *   readObj, getTransformMatrix, buildBVH, buildBVHwithSAH   (P3/main.cpp:254-588)
 *   HDRLoader::load (lib/hdrloader.cpp:50-...)  calculateHdrCache (P5/main.cpp:592-689)
 *   main() run headless (chapter 3 only: its assets are shipped) with the upload calls recorded,
 *   display() run once per camera to capture the `eye` / `cameraRotate` uniforms.
 */
#include <stdio.h>
#include <unistd.h>
#include <cstring>
#include <string>
#include <vector>

#define main ezrt_ref_chapter_main
#include EZRT_REF_MAIN
#undef main

static_assert(sizeof(Triangle) == 144, "reference Triangle is 36 packed floats");
static_assert(sizeof(Triangle_encoded) == 144 && sizeof(BVHNode_encoded) == 48, "reference texel records");

static std::vector<Triangle> g_tris;
static std::vector<BVHNode> g_nodes;

static Material material_from(const float* m18) {
    Material m;
    static_assert(sizeof(Material) == 72, "18 floats");
    std::memcpy(&m, m18, sizeof m);
    return m;
}
static mat4 mat_from(const float* t16) {
    mat4 t;
    std::memcpy(&t, t16, 64);
    return t;
}

extern "C" {

void ref_scene_clear() { g_tris.clear(); g_nodes.clear(); }

void ref_material_default(float* out18) { Material m; std::memcpy(out18, &m, sizeof m); }

void ref_get_transform_matrix(const float* r, const float* t, const float* s, float* out16) {
    mat4 m = getTransformMatrix(vec3(r[0], r[1], r[2]), vec3(t[0], t[1], t[2]), vec3(s[0], s[1], s[2]));
    std::memcpy(out16, &m, 64);
}

void ref_read_obj(const char* path, const float* m18, const float* t16, int smooth) {
    readObj(path, g_tris, material_from(m18), mat_from(t16), smooth != 0);
}

void ref_add_triangles(const float* tri36, int n) {
    size_t o = g_tris.size();
    g_tris.resize(o + n);
    std::memcpy(&g_tris[o], tri36, size_t(n) * 144);
}

/* main()'s build sequence (P3/main.cpp:703-715): dummy node 0, then the builder over everything */
int ref_build(int sah, int leaf_n) {
    BVHNode testNode;
    testNode.left = 255; testNode.right = 128; testNode.n = 30; testNode.index = 0;
    testNode.AA = vec3(1, 1, 0); testNode.BB = vec3(0, 1, 0);
    g_nodes.assign(1, testNode);
    if (sah) return buildBVHwithSAH(g_tris, g_nodes, 0, int(g_tris.size()) - 1, leaf_n);
    return buildBVH(g_tris, g_nodes, 0, int(g_tris.size()) - 1, leaf_n);
}

void ref_counts(int* nt, int* nn) { *nt = int(g_tris.size()); *nn = int(g_nodes.size()); }

/* raw structs: Triangle IS the 36-float record; nodes as (left,right,n,index) + AA + BB */
void ref_get_scene(float* tri36, int* node_ints4, float* node_boxes6) {
    if (!g_tris.empty()) std::memcpy(tri36, g_tris.data(), g_tris.size() * 144);
    for (size_t i = 0; i < g_nodes.size(); i++) {
        node_ints4[4 * i + 0] = g_nodes[i].left; node_ints4[4 * i + 1] = g_nodes[i].right;
        node_ints4[4 * i + 2] = g_nodes[i].n;    node_ints4[4 * i + 3] = g_nodes[i].index;
        std::memcpy(node_boxes6 + 6 * i, &g_nodes[i].AA, 12);
        std::memcpy(node_boxes6 + 6 * i + 3, &g_nodes[i].BB, 12);
    }
}

int ref_hdr_load(const char* path, int* w, int* h, float** cols) {
    HDRLoaderResult res;
    res.width = res.height = 0; res.cols = nullptr;
    bool ok = HDRLoader::load(path, res);
    *w = res.width; *h = res.height; *cols = res.cols;
    return ok ? 1 : 0;
}
void ref_free(float* p) { delete[] p; }

#ifdef EZRT_REF_HAS_HDRCACHE
float* ref_calculate_hdr_cache(float* hdr, int w, int h) { return calculateHdrCache(hdr, w, h); }
#endif

/* display() once with the given mouse state -> the uniforms the shader would see */
void ref_camera(float rotat, float up, float radius, float* eye3, float* cam16) {
    rotatAngle = rotat; upAngle = up; r = radius;
    FILE* keep = stdout;                       /* display() prints an FPS line */
    std::cout.setstate(std::ios_base::failbit);
    display();
    std::cout.clear();
    (void)keep;
    std::memcpy(eye3, g_ezrt_ref_gl.uniforms_f["eye"].data(), 12);
    std::memcpy(cam16, g_ezrt_ref_gl.uniforms_f["cameraRotate"].data(), 64);
}

/* the chapter's main() headless, cwd = its source directory (relative asset paths) */
int ref_run_main(const char* source_dir, int record_images) {
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return -1;
    if (chdir(source_dir) != 0) return -2;
    g_ezrt_ref_gl.clear();
    g_ezrt_ref_gl.record_images = record_images != 0;
    std::cout.setstate(std::ios_base::failbit);
    char arg0[] = "ezrt_ref";
    char* argv[] = {arg0, nullptr};
    int rc = ezrt_ref_chapter_main(1, argv);
    std::cout.clear();
    if (chdir(cwd) != 0) return -3;
    return rc;
}
int ref_recorded_buffers() { return int(g_ezrt_ref_gl.texture_buffers.size()); }
long ref_recorded_buffer_floats(int i) { return long(g_ezrt_ref_gl.texture_buffers[i].data.size()); }
void ref_recorded_buffer_get(int i, float* out) {
    const auto& d = g_ezrt_ref_gl.texture_buffers[i].data;
    std::memcpy(out, d.data(), d.size() * sizeof(float));
}
long long ref_recorded_uniform_i(const char* name) {
    auto it = g_ezrt_ref_gl.uniforms_i.find(name);
    return it == g_ezrt_ref_gl.uniforms_i.end() ? -1 : it->second;
}

}  // extern "C"
