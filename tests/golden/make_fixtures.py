#!/usr/bin/env python3
"""Generate the committed fixtures / data assets from the reference's data files, using the
REFERENCE'S OWN CODE (oracle/_ref, built by oracle/ref_recipe/build_ref.py) wherever an expected
value is frozen.

Run once in the build container (where /root/reference exists):
    python oracle/ref_recipe/build_ref.py && python tests/golden/make_fixtures.py

Writes
  ezrt_amd/assets/meshes.npz
        raw OBJ vertices (float32, parsed with strtof exactly as an OBJ reader would) and 0-based
        faces of the meshes the BASELINE.json configs use: Stanford Bunny (2503 v / 4968 f),
        sphere.obj (320 f), quad.obj (a 12-triangle box), sphere2.obj.  Data assets (the Stanford 3D
        Scanning Repository bunny and Blender exports), not reference source code; the GPU box has no
        /root/reference, so tests and bench.py rebuild OBJ text from these arrays (scenes.obj_text).
  ezrt_amd/assets/env_peppermint_powerplant_1024x512_rgbe.npz
        the RGBE texels (uint8 [512,1024,4], row 0 = top scanline) of the only HDR the reference
        ships, P4/HDR/peppermint_powerplant_4k.hdr (a Poly Haven CC0 panorama), run-length decoded
        here; scenes.shipped_hdr() turns them into floats with HDRLoader's formula
        (lib/hdrloader.cpp:97-114).  Checked below against the reference loader, every texel.
  tests/golden/hdr_probe.json
        size, a few texels, checksums of that map AS DECODED BY THE REFERENCE'S HDRLoader::load and
        of calculateHdrCache's output AS COMPUTED BY THE REFERENCE (P5/main.cpp:592-689).
  tests/golden/ref_scene_p3.json
        what chapter 3's main() uploads when run headless (counts + checksums of both arrays).
"""
import ctypes
import json
import os
import sys

import numpy as np

REF = "/root/reference"
P3 = os.path.join(REF, "part 3 -- OpenGL Raytracing", "source code")
P4 = os.path.join(REF, "part 4 -- Disney Principle BRDF", "source code")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ASSETS = os.path.join(ROOT, "ezrt_amd", "assets")
sys.path.insert(0, ROOT)

_libc = ctypes.CDLL("libc.so.6")
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]


def parse_obj(path):
    verts, faces = [], []
    with open(path, "rb") as f:
        for line in f.read().split(b"\n"):
            tok = line.split()
            if not tok:
                continue
            if tok[0] == b"v":
                verts.append([_libc.strtof(t, None) for t in tok[1:4]])
            elif tok[0] == b"f":
                faces.append([int(t.split(b"/")[0]) - 1 for t in tok[1:4]])
    return np.array(verts, np.float32), np.array(faces, np.int32)


def rgbe_texels(path):
    """Radiance .hdr -> uint8 [h, w, 4] (new-style RLE scanlines only, which is what the file uses)."""
    raw = open(path, "rb").read()
    assert raw[:10] == b"#?RADIANCE"
    p = raw.index(b"\n\n") + 2
    q = raw.index(b"\n", p)
    tok = raw[p:q].split()
    assert tok[0] == b"-Y" and tok[2] == b"+X"
    h, w = int(tok[1]), int(tok[3])
    p = q + 1
    out = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        assert raw[p] == 2 and raw[p + 1] == 2 and ((raw[p + 2] << 8) | raw[p + 3]) == w
        p += 4
        for c in range(4):
            x = 0
            while x < w:
                code = raw[p]
                p += 1
                if code > 128:
                    n = code & 127
                    out[y, x:x + n, c] = raw[p]
                    p += 1
                else:
                    n = code
                    out[y, x:x + n, c] = np.frombuffer(raw, np.uint8, n, p)
                    p += n
                x += n
    return out


def bits_xor(a):
    return int(np.bitwise_xor.reduce(np.ascontiguousarray(a, np.float32).view(np.uint32).ravel()))


def bits_sum(a):
    return int(np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64).sum() & np.uint64(0xFFFFFFFFFFFFFFFF))


def main():
    from oracle import ref as R
    os.makedirs(ASSETS, exist_ok=True)
    out = {}
    for name, fn in (("bunny", "Stanford Bunny.obj"), ("sphere", "sphere.obj"), ("quad", "quad.obj"),
                     ("sphere2", "sphere2.obj")):
        v, f = parse_obj(os.path.join(P3, "models", fn))
        print(name, v.shape, f.shape)
        out[name + "_v"] = v
        out[name + "_f"] = f
    np.savez_compressed(os.path.join(ASSETS, "meshes.npz"), **out)

    hdr_path = os.path.join(P4, "HDR", "peppermint_powerplant_4k.hdr")
    ref5 = R.Flat("p5")
    hdr = ref5.hdrLoad(hdr_path)                       # the reference's HDRLoader::load
    rgbe = rgbe_texels(hdr_path)
    np.savez_compressed(os.path.join(ASSETS, "env_peppermint_powerplant_1024x512_rgbe.npz"), rgbe=rgbe)
    from ezrt_amd import scenes
    scenes._shipped_hdr_cache.clear()
    assert np.array_equal(scenes.shipped_hdr().view(np.uint32), hdr.view(np.uint32)), "asset != reference decode"
    cache = ref5.calculateHdrCache(hdr)                # the reference's calculateHdrCache
    h, w, _ = hdr.shape
    probes = [(0, 0), (h // 2, w // 2), (h - 1, w - 1), (100, 700), (300, 123), (17, 1000), (480, 5)]
    info = {
        "file": "part 4 -- Disney Principle BRDF/source code/HDR/peppermint_powerplant_4k.hdr",
        "produced_by": "oracle/_ref (reference HDRLoader::load + calculateHdrCache), tests/golden/make_fixtures.py",
        "width": w, "height": h,
        "probes": [{"row": r, "col": c, "rgb_bits": [int(x) for x in hdr[r, c].view(np.uint32)],
                    "cache_bits": [int(x) for x in cache[r, c].view(np.uint32)]} for r, c in probes],
        "sum_f64": float(hdr.astype(np.float64).sum()),
        "max": float(hdr.max()),
        "xor_bits": bits_xor(hdr), "sum_bits": bits_sum(hdr),
        "cache_xor_bits": bits_xor(cache), "cache_sum_bits": bits_sum(cache),
    }
    with open(os.path.join(HERE, "hdr_probe.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(info["width"], info["height"], info["sum_f64"], info["max"])

    ref3 = R.Flat("p3")
    bufs = ref3.runMain()                              # chapter 3's main(), headless
    tri, nodes = bufs[0].reshape(-1, 36), bufs[1].reshape(-1, 12)
    scene = {
        "produced_by": "oracle/_ref: P3/main.cpp main() run headless, glBufferData(GL_TEXTURE_BUFFER) payloads",
        "nTriangles": int(tri.shape[0]), "nNodes": int(nodes.shape[0]),
        "tri_xor_bits": bits_xor(tri), "tri_sum_bits": bits_sum(tri),
        "nodes_xor_bits": bits_xor(nodes), "nodes_sum_bits": bits_sum(nodes),
        "node1": [float(x) for x in nodes[1]], "tri0": [float(x) for x in tri[0]],
    }
    with open(os.path.join(HERE, "ref_scene_p3.json"), "w") as f:
        json.dump(scene, f, indent=1)
    print(scene["nTriangles"], scene["nNodes"])


if __name__ == "__main__":
    main()
