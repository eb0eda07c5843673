/* TEST INFRASTRUCTURE -- oracle/_ref/libezrt_ref_fsh_p{3,4,5}.so: the reference's FRAGMENT SHADER, compiled.
 *
 * This translation unit #includes the text of one chapter's `shaders/fshader.fsh` -- read from /root/reference at
 * build time and passed through the syntax-only source pass of fsh_pass.py (EZRT_REF_FSH = the pass' temporary
 * output; nothing of it is kept) -- as the body of `struct Fsh`, with shim/glsl_shim.h supplying the GLSL language.
 * One Fsh object = one fragment invocation: its default member initialisers are the shader's file-scope
 * initialisers (`uint seed = ...`), `main()` is the shader's main().  The exports below run it per pixel-sample
 * and give function-level access to the shader's own BRDF / sampler / pdf / traversal functions, so that
 * oracle/ezrt_oracle.c's restatement of P5/fsh:160-890 (P4/fsh:412-517, P3/fsh:376-413) is checked by EXECUTION
 * (tests/test_ref_fsh_pin.py), not by reading.
 *
 * Build: oracle/ref_recipe/build_ref.py (g++ -O2 -ffp-contract=off -fno-fast-math -fsingle-precision-constant).
 */
#include <cstring>
#include <vector>

#include "glsl_shim.h"

#ifndef EZRT_FSH_CHAPTER
#error "EZRT_FSH_CHAPTER must be 3, 4 or 5"
#endif

struct Fsh : GlslBuiltins {
  inline static int ezrt_max_bounce = 2; /* fsh_pass.py rule 5 */
  inline static int ezrt_use_is = 1;
  vec4 gl_FragData[4];
#include EZRT_REF_FSH
};

namespace {

std::vector<float> g_tri, g_nodes, g_hdr, g_cache;
float g_last[3] = {0.0f, 0.0f, 0.0f};
int g_w = 1, g_h = 1;

void set_pixel(int ix, int iy, unsigned frame) {
  /* the full-screen quad's interpolated position at the pixel centre (pix = ((i+.5)/W)*2-1) */
  Fsh::pix = vec3(((float)ix + 0.5f) / (float)g_w * 2.0f - 1.0f, ((float)iy + 0.5f) / (float)g_h * 2.0f - 1.0f, 0.0f);
  Fsh::frameCounter = frame;
}

Fsh::Material material_from(const float* m) {
  Fsh::Material r;
  r.emissive = vec3(m[0], m[1], m[2]);
  r.baseColor = vec3(m[3], m[4], m[5]);
  r.subsurface = m[6];
  r.metallic = m[7];
  r.specular = m[8];
  r.specularTint = m[9];
  r.roughness = m[10];
  r.anisotropic = m[11];
  r.sheen = m[12];
  r.sheenTint = m[13];
  r.clearcoat = m[14];
  r.clearcoatGloss = m[15];
  r.IOR = m[16];
  r.transmission = m[17];
  return r;
}
vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }
void put(float* o, const vec3& v) {
  o[0] = v.x;
  o[1] = v.y;
  o[2] = v.z;
}

} // namespace

extern "C" {

int fsh_chapter(void) { return EZRT_FSH_CHAPTER; }

/* glBufferData of the two buffer textures (P5/main.cpp:878-893) */
void fsh_set_scene(const float* tri, int n_tri, const float* nodes, int n_nodes) {
  g_tri.assign(tri, tri + (size_t)n_tri * 36);
  g_nodes.assign(nodes, nodes + (size_t)n_nodes * 12);
  Fsh::triangles.data = g_tri.data();
  Fsh::nodes.data = g_nodes.data();
  Fsh::nTriangles = n_tri;
  Fsh::nNodes = n_nodes;
}

/* glTexImage2D of hdrMap / hdrCache (P5/main.cpp:896-906); bilinear = the GL_LINEAR of chapter 5 (P5/main.cpp:196-197) */
void fsh_set_env(const float* hdr, const float* cache, int w, int h, int bilinear) {
  g_hdr.assign(hdr, hdr + (size_t)w * h * 3);
  Fsh::hdrMap.data = g_hdr.data();
  Fsh::hdrMap.W = w;
  Fsh::hdrMap.H = h;
  Fsh::hdrMap.bilinear = bilinear;
#if EZRT_FSH_CHAPTER == 5
  if (cache) {
    g_cache.assign(cache, cache + (size_t)w * h * 3);
    Fsh::hdrCache.data = g_cache.data();
    Fsh::hdrCache.W = w;
    Fsh::hdrCache.H = h;
    Fsh::hdrCache.bilinear = bilinear;
  }
  Fsh::hdrResolution = w;
#else
  (void)cache;
#endif
}

/* the uniforms of display() (P5/main.cpp:717-720) */
void fsh_set_camera(const float* eye, const float* camera_rotate16, int width, int height) {
  Fsh::eye = vec3(eye[0], eye[1], eye[2]);
  std::memcpy(Fsh::cameraRotate.m, camera_rotate16, 64);
  Fsh::width = width;
  Fsh::height = height;
  g_w = width;
  g_h = height;
  /* lastFrame: the previous running mean of the pixel being shaded, as a 1x1 NEAREST texture */
  Fsh::lastFrame.data = g_last;
  Fsh::lastFrame.W = 1;
  Fsh::lastFrame.H = 1;
  Fsh::lastFrame.bilinear = 0;
}

void fsh_set_integrator(int max_bounce, int use_importance_sampling) {
  Fsh::ezrt_max_bounce = max_bounce;
  Fsh::ezrt_use_is = use_importance_sampling;
}

/* frames [frame0, frame0 + spp) of the pixel rect through the shader's main(): accum RGBA32F [height][width][4], row 0 =
 * bottom, in = lastFrame after frame0 frames, out = after frame0 + spp (the mix is main()'s own, P5/fsh:943-947) */
void fsh_render(int x0, int y0, int x1, int y1, unsigned frame0, unsigned spp, float* accum) {
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++) {
      float* px = accum + ((size_t)y * g_w + x) * 4;
      for (unsigned k = 0; k < spp; k++) {
        const unsigned frame = frame0 + k;
        g_last[0] = frame == 0 ? 0.0f : px[0]; /* frame 0: a cleared target (weight of lastFrame is exactly 0) */
        g_last[1] = frame == 0 ? 0.0f : px[1];
        g_last[2] = frame == 0 ? 0.0f : px[2];
        set_pixel(x, y, frame);
        Fsh f;
        f.main();
        px[0] = f.gl_FragData[0].x;
        px[1] = f.gl_FragData[0].y;
        px[2] = f.gl_FragData[0].z;
        px[3] = f.gl_FragData[0].w;
      }
    }
}

/* the seed a fragment at (ix, iy, frame) starts from (P5/fsh:315-318): checks pix -> integer pixel */
unsigned fsh_seed(int ix, int iy, unsigned frame) {
  set_pixel(ix, iy, frame);
  Fsh f;
  return f.seed;
}

/* function-level access; the same op numbers as ezrt_oracle_fn (oracle/ezrt_oracle.c).  Returns 0, or -1 for an op the
 * chapter's shader does not have. */
int fsh_fn(int op, const float* a, const float* b, int n, float* out) {
  set_pixel(0, 0, 0);
  for (int i = 0; i < n; i++) {
    Fsh f;
    switch (op) {
      case 1: { /* BRDF_Evaluate, isotropic (chapter 5): a = V N L, b = material */
#if EZRT_FSH_CHAPTER == 5
        put(out + 3 * i, f.BRDF_Evaluate(v3(a + 9 * i), v3(a + 9 * i + 3), v3(a + 9 * i + 6), material_from(b + 18 * i)));
        break;
#else
        return -1;
#endif
      }
      case 2: { /* the evaluate of the uniform-sampling loop with X, Y = getTangent(N) as that loop calls it:
                   chapter 4 BRDF_Evaluate (anisotropic, P4/fsh:412), chapter 5 BRDF_Evaluate_aniso (P5/fsh:437) */
#if EZRT_FSH_CHAPTER >= 4
        vec3 X, Y;
        const vec3 N = v3(a + 9 * i + 3);
        f.getTangent(N, X, Y);
#if EZRT_FSH_CHAPTER == 5
        put(out + 3 * i, f.BRDF_Evaluate_aniso(v3(a + 9 * i), N, v3(a + 9 * i + 6), X, Y, material_from(b + 18 * i)));
#else
        put(out + 3 * i, f.BRDF_Evaluate(v3(a + 9 * i), N, v3(a + 9 * i + 6), X, Y, material_from(b + 18 * i)));
#endif
        break;
#else
        return -1;
#endif
      }
#if EZRT_FSH_CHAPTER == 5
      case 3: /* SampleBRDF: a = xi1 xi2 xi3 V N */
        put(out + 3 * i, f.SampleBRDF(a[9 * i], a[9 * i + 1], a[9 * i + 2], v3(a + 9 * i + 3), v3(a + 9 * i + 6), material_from(b + 18 * i)));
        break;
      case 4: /* BRDF_Pdf: a = V N L */
        out[i] = f.BRDF_Pdf(v3(a + 9 * i), v3(a + 9 * i + 3), v3(a + 9 * i + 6), material_from(b + 18 * i));
        break;
      case 5: /* hdrPdf: a = L */
        out[i] = f.hdrPdf(v3(a + 3 * i), Fsh::hdrResolution);
        break;
      case 6: /* SampleHdr: a = xi1 xi2 */
        put(out + 3 * i, f.SampleHdr(a[2 * i], a[2 * i + 1]));
        break;
      case 7: /* hdrColor: a = L */
        put(out + 3 * i, f.hdrColor(v3(a + 3 * i)));
        break;
      case 9: /* toNormalHemisphere(SampleHemisphere(xi1, xi2), N): a = xi1 xi2 N */
        put(out + 3 * i, f.toNormalHemisphere(f.SampleHemisphere(a[5 * i], a[5 * i + 1]), v3(a + 5 * i + 2)));
        break;
#else
      case 7: /* sampleHdr (chapter 3 clamps to 10, P3/fsh:154) */
        put(out + 3 * i, f.sampleHdr(v3(a + 3 * i)));
        break;
#endif
      case 8: { /* hitBVH: a = S d; out = isHit isInside distance hitPoint normal baseColor */
        Fsh::Ray r;
        r.startPoint = v3(a + 6 * i);
        r.direction = v3(a + 6 * i + 3);
        const Fsh::HitResult h = f.hitBVH(r);
        float* o = out + 12 * i;
        for (int k = 0; k < 12; k++) o[k] = 0.0f;
        o[0] = h.isHit ? 1.0f : 0.0f;
        o[2] = h.distance;
        if (h.isHit) {
          o[1] = h.isInside ? 1.0f : 0.0f;
          put(o + 3, h.hitPoint);
          put(o + 6, h.normal);
          put(o + 9, h.material.baseColor);
        }
        break;
      }
      default: return -1;
    }
  }
  return 0;
}

} // extern "C"
