#!/usr/bin/env python3
"""Generate the committed fixtures from the reference's data assets.

Run once in the build container (where /root/reference exists):
    python tests/golden/make_fixtures.py

Writes
  tests/golden/meshes.npz   raw OBJ vertices (float32, parsed with strtof exactly as an
                            OBJ reader would) and 0-based faces of the meshes the
                            BASELINE.json configs use: Stanford Bunny (2503 v / 4968 f),
                            sphere.obj (320 f), quad.obj (a 12-triangle box).  These are
                            data assets (the Stanford 3D Scanning Repository bunny and two
                            Blender/MagicaVoxel exports), not reference source code; the GPU
                            box has no /root/reference so tests and bench.py rebuild OBJ text
                            from these arrays (ezrt_amd/scenes.py: obj_text).
  tests/golden/hdr_probe.json  size + a few decoded texels + checksums of the only shipped
                            HDR (P4/HDR/peppermint_powerplant_4k.hdr, 1024x512) decoded by
                            our HDRLoader, used by tests/test_hdr.py when the file is present.
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
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

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


def main():
    out = {}
    for name, fn in (("bunny", "Stanford Bunny.obj"), ("sphere", "sphere.obj"), ("quad", "quad.obj")):
        v, f = parse_obj(os.path.join(P3, "models", fn))
        print(name, v.shape, f.shape)
        out[name + "_v"] = v
        out[name + "_f"] = f
    np.savez_compressed(os.path.join(HERE, "meshes.npz"), **out)

    from ezrt_amd import scene as S
    hdr = S.hdrLoad(os.path.join(P4, "HDR", "peppermint_powerplant_4k.hdr"))
    h, w, _ = hdr.shape
    probes = [(0, 0), (h // 2, w // 2), (h - 1, w - 1), (100, 700), (300, 123)]
    info = {
        "file": "part 4 -- Disney Principle BRDF/source code/HDR/peppermint_powerplant_4k.hdr",
        "width": w, "height": h,
        "probes": [{"row": r, "col": c, "rgb_bits": [int(x) for x in hdr[r, c].view(np.uint32)]} for r, c in probes],
        "sum_f64": float(hdr.astype(np.float64).sum()),
        "max": float(hdr.max()),
        "xor_bits": int(np.bitwise_xor.reduce(hdr.view(np.uint32).ravel())),
    }
    with open(os.path.join(HERE, "hdr_probe.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(info["width"], info["height"], info["sum_f64"], info["max"])


if __name__ == "__main__":
    main()
