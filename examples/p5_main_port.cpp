// p5_main_port.cpp -- chapter 5's main() (P5/main.cpp:763-947) and display() (P5/main.cpp:697-748)
// with the window, the GL objects and the shaders taken out and this repo's libraries put in:
//
//   what main() did                                              here
//   readObj x 2, testNode, buildBVHwithSAH   P5/main.cpp:795-838   ezrt::readObj / ezrt::testNode / ezrt::buildBVHwithSAH (same calls)
//   the two encode loops                     P5/main.cpp:841-871   ezrt::encodeTriangles / ezrt::encodeBVH
//   tbo0 / tbo1 + glTexBuffer                P5/main.cpp:878-893   ezrt_scene_create
//   HDRLoader::load + calculateHdrCache      P5/main.cpp:896-906   ezrt::HDRLoader::load / ezrt::calculateHdrCache -> ezrt_scene_set_env
//   eye / cameraRotate / frameCounter        P5/main.cpp:710-720   EzrtRenderParams
//   pass1.draw(); pass2.draw() per frame     P5/main.cpp:743-744   ezrt_render (spp frames per call, lastFrame = accum)
//   pass3.draw()                             P5/main.cpp:745       ezrt_tonemap
//
// It is the INTEGRATION.md section 2 stub made real: the only translation unit outside the libraries that
// includes ezrt.h and ezrt_scene.hpp, built by `make examples` (g++, no hipcc: the C ABI needs no HIP header)
// and run by tests/test_gpu_example_port.py, which diffs its PFM against the Python binding's frame.
//
// usage: p5_main_port <model.obj> <quad.obj> <env.hdr> <out.pfm> [size=512] [frames=16] [out.ppm]
//                     [tx ty tz scale]   (model transform; default = the teapot's: 0 -0.5 0 0.75)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "ezrt.h"
#include "ezrt_scene.hpp"

using namespace ezrt;

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cerr << "usage: " << argv[0] << " model.obj quad.obj env.hdr out.pfm [size] [frames] [out.ppm] [tx ty tz scale]\n";
    return 2;
  }
  const int size = argc > 5 ? atoi(argv[5]) : 512;
  const unsigned frames = argc > 6 ? (unsigned)atoi(argv[6]) : 16u;
  const char* ppm = argc > 7 && argv[7][0] ? argv[7] : nullptr;
  vec3 mt(0, -0.5f, 0);
  float ms = 0.75f;
  if (argc > 11) {
    mt = vec3((float)atof(argv[8]), (float)atof(argv[9]), (float)atof(argv[10]));
    ms = (float)atof(argv[11]);
  }

  // camera config (P5/main.cpp:775-777)
  const float rotatAngle = 90, upAngle = 10, r = 2.0f;

  // scene config (P5/main.cpp:780-819): P4/P5 Material defaults, then the fields main() sets
  std::vector<Triangle> triangles;
  Material m = disneyDefaults();
  m.roughness = 0.5f;
  m.specular = 1.0f;
  m.metallic = 1.0f;
  m.clearcoat = 1.0f;
  m.clearcoatGloss = 0.0f;
  m.baseColor = vec3(1, 0.73f, 0.25f);
  try {
    readObj(argv[1], triangles, m, getTransformMatrix(vec3(0, 0, 0), mt, vec3(ms, ms, ms)), true);
    m.roughness = 0.01f;
    m.metallic = 0.1f;
    m.specular = 1.0f;
    m.baseColor = vec3(1, 1, 1);
    const float len = 13000.0f;
    readObj(argv[2], triangles, m, getTransformMatrix(vec3(0, 0, 0), vec3(0, -0.5f, 0), vec3(len, 0.01f, len)), false);
  } catch (const std::runtime_error& e) { // (the reference: cout + exit(-1), P3/main.cpp:282-285)
    std::cerr << e.what() << std::endl;
    return 1;
  }
  const int nTriangles = (int)triangles.size();
  std::cout << "triangles: " << nTriangles << std::endl;

  // BVH (P5/main.cpp:824-838)
  std::vector<BVHNode> nodes{testNode()};
  buildBVHwithSAH(triangles, nodes, 0, nTriangles - 1, 8);
  const int nNodes = (int)nodes.size();
  std::cout << "BVH nodes: " << nNodes << std::endl;

  // encode (P5/main.cpp:841-871)
  std::vector<Triangle_encoded> triangles_encoded = encodeTriangles(triangles);
  std::vector<BVHNode_encoded> nodes_encoded = encodeBVH(nodes);

  // upload (P5/main.cpp:878-893)
  EzrtScene* scene = nullptr;
  if (ezrt_scene_create(&triangles_encoded[0].p1.x, nTriangles, &nodes_encoded[0].childs.x, nNodes, &scene)) {
    std::cerr << "ezrt_scene_create: " << ezrt_last_error() << std::endl;
    return 1;
  }

  // env map + importance-sampling cache (P5/main.cpp:896-906)
  HDRLoaderResult hdrRes;
  if (!HDRLoader::load(argv[3], hdrRes)) { // (the reference ignores the bool)
    std::cerr << "cannot load " << argv[3] << std::endl;
    return 1;
  }
  float* cache = calculateHdrCache(hdrRes.cols, hdrRes.width, hdrRes.height);
  if (ezrt_scene_set_env(scene, hdrRes.cols, cache, hdrRes.width, hdrRes.height, EZRT_FILTER_BILINEAR)) {
    std::cerr << "ezrt_scene_set_env: " << ezrt_last_error() << std::endl;
    return 1;
  }
  delete[] cache;
  delete[] hdrRes.cols;

  // display() x frames (P5/main.cpp:697-748): uniforms -> params, pass1 + pass2 -> ezrt_render
  const Camera cam = cameraFromAngles(rotatAngle, upAngle, r); // lines 710-713
  EzrtRenderParams rp;
  memset(&rp, 0, sizeof rp);
  rp.width = rp.height = size;
  rp.x1 = rp.y1 = size;
  rp.max_bounce = 2; // fshader.fsh:935
  rp.integrator = EZRT_INTEGRATOR_P5_MIS; // fshader.fsh:936 (pathTracingImportanceSampling)
  memcpy(rp.eye, &cam.eye.x, sizeof rp.eye);
  memcpy(rp.camera_rotate, &cam.cameraRotate.c[0][0], sizeof rp.camera_rotate);
  std::vector<float> lastFrame((size_t)size * size * 4, 0.0f);
  const unsigned per_call = 8; // display() draws one frame per call; a batch of them is the same running mean
  for (unsigned frameCounter = 0; frameCounter < frames; frameCounter += rp.spp) {
    rp.frame0 = frameCounter;
    rp.spp = frames - frameCounter < per_call ? frames - frameCounter : per_call;
    if (ezrt_render(scene, &rp, lastFrame.data())) {
      std::cerr << "ezrt_render: " << ezrt_last_error() << std::endl;
      return 1;
    }
  }
  uint64_t ctr[EZRT_CTR_COUNT];
  ezrt_counters(scene, ctr);
  std::cout << "backend " << ezrt_backend() << ", rays " << (unsigned long long)ctr[EZRT_CTR_RAYS] << ", samples "
            << (unsigned long long)ctr[EZRT_CTR_SAMPLES] << std::endl;

  // lastFrame as a PFM (rows bottom first, like the GL target)
  FILE* f = fopen(argv[4], "wb");
  if (!f) return 1;
  fprintf(f, "PF\n%d %d\n-1.0\n", size, size);
  for (size_t i = 0; i < (size_t)size * size; i++) fwrite(&lastFrame[i * 4], sizeof(float), 3, f);
  fclose(f);

  if (ppm) { // pass3 (tone map + gamma) + chapter 1's 8-bit quantisation, top row first
    std::vector<uint8_t> rgb8((size_t)size * size * 3);
    if (ezrt_tonemap(lastFrame.data(), size * size, rgb8.data())) {
      std::cerr << "ezrt_tonemap: " << ezrt_last_error() << std::endl;
      return 1;
    }
    FILE* g = fopen(ppm, "wb");
    if (!g) return 1;
    fprintf(g, "P6\n%d %d\n255\n", size, size);
    for (int y = size - 1; y >= 0; y--) fwrite(&rgb8[(size_t)y * size * 3], 1, (size_t)size * 3, g);
    fclose(g);
  }
  ezrt_scene_destroy(scene);
  return 0;
}
