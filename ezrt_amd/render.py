"""Scene description file -> image: what each chapter's hard-coded main() does
(P3/main.cpp:676-748 ... P5/main.cpp:784-947), driven by a JSON file instead of being compiled in.

    python -m ezrt_amd.render scene.json -o out.png [--pfm out.pfm] [--spp N] [--checkpoint ck.npz]

{
  "width": 512, "height": 512, "spp": 64, "max_bounce": 4,
  "integrator": 50,                      # 3 | 4 | 50 | 51 (ezrt.h)
  "camera": {"rotatAngle": 0, "upAngle": 0, "r": 4},
  "env": {"hdr": "sky.hdr", "filter": "bilinear", "clamp": 0},        # or {"synthetic": true}
  "bvh": {"builder": "sah", "leaf": 8},                  # sah (GPU builder from 100 000 triangles on, else host: same tree) |
                                                         # sah_gpu | sah_host (force one) | median | lbvh (GPU, another tree)
  "objects": [
    {"obj": "bunny.obj", "smooth": true,
     "rotate": [0, 0, 0], "translate": [0.3, -1.6, 0], "scale": [1.5, 1.5, 1.5],
     "material": {"defaults": "p4", "baseColor": [1, 1, 1], "roughness": 0.5}},
    {"triangles": [[[1, -1, 1], [-1, -1, -1], [-1, -1, 1]]], "material": {"baseColor": [1, 1, 1]}}   # inline triangles (part 1's Cornell box)
  ]
}
Paths are relative to the scene file.  Material keys are the fields of struct Material
(P3/main.cpp:28-43); "defaults": "p3" | "p4" picks the chapter's default values.  `.mtl` files are
ignored, as in the reference."""
import argparse
import json
import os
import sys
import time


from . import build, imageio, progressive, scenes, trace
from . import scene as S
from ._abi import FILTER_BILINEAR, FILTER_NEAREST


def material_from(spec):
    spec = dict(spec or {})
    which = spec.pop("defaults", "p4")
    mk = S.Material if which == "p3" else S.Material.disney
    return mk(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in spec.items()})


def inline_triangles(tris, material):
    """[[p1, p2, p3], ...] -> [n, 36] records in the reference's Triangle layout (P3/main.cpp:61-72): flat normal
    normalize(cross(p2 - p1, p3 - p1)) in float32 at all three vertices, as part 1's Triangle constructor computes it."""
    import numpy as np
    out = []
    m18 = material.to18()
    for t in tris:
        p1, p2, p3 = (np.array(p, np.float32) for p in t)
        n = np.cross(p2 - p1, p3 - p1).astype(np.float32)
        n = n / np.float32(np.sqrt(np.float32(n @ n)))
        out.append(np.concatenate([p1, p2, p3, n, n, n, m18]).astype(np.float32))
    return np.stack(out)


def build_scene(desc, base_dir="."):
    """-> scenes.BuiltScene (host arrays in the reference layouts), following main()'s sequence:
    readObj per object, nodes = {testNode}, buildBVHwithSAH / buildBVH, encode."""
    hs = S.HostScene()
    for o in desc["objects"]:
        if "triangles" in o:   # an inline triangle list (part 1's hard-coded Cornell box, P1/main.cpp:338-360): no OBJ, no normalisation
            hs.addTriangles(inline_triangles(o["triangles"], material_from(o.get("material"))))
            continue
        trans = S.getTransformMatrix(tuple(o.get("rotate", (0, 0, 0))), tuple(o.get("translate", (0, 0, 0))),
                                     tuple(o.get("scale", (1, 1, 1))))
        hs.readObj(os.path.join(base_dir, o["obj"]), material_from(o.get("material")), trans, bool(o.get("smooth", False)))
    bvh = desc.get("bvh", {})
    builder, leaf = bvh.get("builder", "sah"), int(bvh.get("leaf", 8))
    env = desc.get("env") or {}
    hdr = None
    if env.get("synthetic"):
        hdr = scenes.synthetic_hdr()
    elif env.get("hdr"):
        hdr = S.hdrLoad(os.path.join(base_dir, env["hdr"]))
    want_cache = int(desc.get("integrator", 50)) == 51
    filt = FILTER_NEAREST if env.get("filter", "bilinear") == "nearest" else FILTER_BILINEAR
    if builder == "lbvh":
        tri, _ = hs.encode()  # triangles in readObj order; the GPU builder orders them itself
        tri, nodes, ms = build.build_lbvh(tri, leaf)
        cache = S.calculateHdrCache(hdr) if (want_cache and hdr is not None) else None
        return scenes.BuiltScene("file", tri, nodes, {"lbvh_ms": ms}, hdr, cache, filt)
    return scenes._finish("file", hs, leaf, hdr, want_cache, filt, sah=(builder != "median"),
                          gpu_build=True if builder == "sah_gpu" else (False if builder == "sah_host" else None))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("scene")
    ap.add_argument("-o", "--png", default="")
    ap.add_argument("--pfm", default="")
    ap.add_argument("--spp", type=int, default=0, help="override the file's spp")
    ap.add_argument("--checkpoint", default="", help="resume from / save to this .npz")
    a = ap.parse_args(argv)
    desc = json.load(open(a.scene))
    t0 = time.time()
    built = build_scene(desc, os.path.dirname(os.path.abspath(a.scene)))
    t1 = time.time()
    gpu = built.upload(trace.hip())
    cam = desc.get("camera", {})
    env = desc.get("env") or {}
    kw = dict(width=desc.get("width", 512), height=desc.get("height", 512), integrator=desc.get("integrator", 50),
              max_bounce=desc.get("max_bounce", 4), env_clamp=float(env.get("clamp", 0.0)),
              rotatAngle=cam.get("rotatAngle", 0.0), upAngle=cam.get("upAngle", 0.0), r=cam.get("r", 4.0))
    if a.checkpoint and os.path.exists(a.checkpoint):
        pr = progressive.ProgressiveRenderer.load(a.checkpoint, gpu)
    else:
        pr = progressive.ProgressiveRenderer(gpu, **kw)
    spp = a.spp or int(desc.get("spp", 64))
    todo = max(0, spp - pr.frameCounter)
    t2 = time.time()
    if todo:
        pr.step(todo)
    t3 = time.time()
    if a.checkpoint:
        pr.save(a.checkpoint)
    if a.pfm:
        imageio.write_pfm(a.pfm, pr.accum)
    if a.png:
        rgb8 = trace.hip().tonemap(pr.accum.reshape(-1, 4)).reshape(pr.height, pr.width, 3)   # pass3.fsh + 8-bit
        imageio.write_png(a.png, rgb8)
    print("scene: %d triangles, %d nodes, build %.2f s; %d spp in %.3f s" %
          (built.tri.shape[0], built.nodes.shape[0], t1 - t0, todo, t3 - t2), file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
