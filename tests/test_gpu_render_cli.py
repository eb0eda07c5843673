"""Scene description file -> image (ezrt_amd/render.py): the P3 scene written as JSON
must give exactly the frame of the canned scene, through the checkpoint path as well."""
import json

import numpy as np
import pytest

from ezrt_amd import imageio, render, scenes, trace
from ezrt_amd import scene as S

pytestmark = pytest.mark.gpu


def test_scene_file_reproduces_the_canned_p3_scene(hip, bunny_small, tmp_path):
    for name in ("bunny", "quad", "sphere"):
        v, f = scenes.mesh(name)
        (tmp_path / (name + ".obj")).write_bytes(scenes.obj_text(v, f))
    desc = {
        "width": 96, "height": 64, "spp": 6, "max_bounce": 4, "integrator": 50,
        "camera": {"rotatAngle": 0, "upAngle": 0, "r": 4},
        "env": {"synthetic": True, "filter": "bilinear"},
        "bvh": {"builder": "sah", "leaf": 8},
        "objects": [
            {"obj": "bunny.obj", "smooth": True, "translate": [0.3, -1.6, 0], "scale": [1.5, 1.5, 1.5],
             "material": {"defaults": "p4", "baseColor": [1, 1, 1]}},
            {"obj": "quad.obj", "translate": [0, -1.4, 0], "scale": [18.83, 0.01, 18.83],
             "material": {"defaults": "p4", "baseColor": [0.725, 0.71, 0.68]}},
            {"obj": "sphere.obj", "translate": [0.0, 0.9, 0.0],
             "material": {"defaults": "p4", "baseColor": [1, 1, 1], "emissive": [30, 20, 10]}},
        ],
    }
    (tmp_path / "scene.json").write_text(json.dumps(desc))
    built = render.build_scene(desc, str(tmp_path))
    assert np.array_equal(built.tri, bunny_small.tri) and np.array_equal(built.nodes, bunny_small.nodes)
    # 4 samples, checkpoint, then the remaining 2 in a second invocation
    args = [str(tmp_path / "scene.json"), "--checkpoint", str(tmp_path / "ck.npz")]
    assert render.main(args + ["--spp", "4"]) == 0
    assert render.main(args + ["-o", str(tmp_path / "out.png"), "--pfm", str(tmp_path / "out.pfm")]) == 0
    eye, cam = S.camera(0, 0, 4)
    want = bunny_small.upload(hip).render(trace.make_params(96, 64, eye, cam, 50, 4, spp=6))
    got = imageio.read_pfm(tmp_path / "out.pfm")
    assert np.array_equal(got.view(np.uint32), want[..., :3].view(np.uint32))
    assert (tmp_path / "out.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    # the GPU builder as an option of the same file
    desc["bvh"] = {"builder": "lbvh", "leaf": 8}
    lb = render.build_scene(desc, str(tmp_path))
    assert lb.tri.shape == bunny_small.tri.shape and lb.nodes.shape[0] > 2
    img = lb.upload(hip).render(trace.make_params(96, 64, eye, cam, 50, 4, spp=6))
    assert np.abs(img[..., :3] - want[..., :3]).max() < 1e-3     # same scene, different tree: only exact-tie order could differ


def _read_png_rgb8(path):
    """decode the truecolour, unfiltered PNG imageio.write_png writes (svpng's form, P1/svpng.inc) -> uint8 [H, W, 3], top row first"""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, tag = struct.unpack(">I", b[pos:pos + 4])[0], b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + data) & 0xFFFFFFFF == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", data[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_c1_cornell_box_as_stated_through_the_scene_file(hip, oracle, tmp_path):
    """BASELINE.json configs[0] end to end: part 1's 12-triangle Cornell box (P1/main.cpp:338-360) as a scene
    file of inline triangles, 256x256, 1 spp, written as PNG by the CLI -- and the PNG's pixels are the tone-mapped 8-bit
    image of the oracle's frame for CONFIGS["C1"], the frame itself equal on the bits."""
    cfg = scenes.CONFIGS["C1"]
    RED, GREEN, WHITE = [1, 0.5, 0.5], [0.5, 1, 0.5], [1, 1, 1]
    light = [[[0.4, 0.99, 0.4], [-0.4, 0.99, -0.4], [-0.4, 0.99, 0.4]], [[0.4, 0.99, 0.4], [0.4, 0.99, -0.4], [-0.4, 0.99, -0.4]]]
    white = [[[1, -1, 1], [-1, -1, -1], [-1, -1, 1]], [[1, -1, 1], [1, -1, -1], [-1, -1, -1]],
             [[1, 1, 1], [-1, 1, 1], [-1, 1, -1]], [[1, 1, 1], [-1, 1, -1], [1, 1, -1]],
             [[1, -1, -1], [-1, 1, -1], [-1, -1, -1]], [[1, -1, -1], [1, 1, -1], [-1, 1, -1]]]
    red = [[[-1, -1, -1], [-1, 1, 1], [-1, -1, 1]], [[-1, -1, -1], [-1, 1, -1], [-1, 1, 1]]]
    green = [[[1, 1, 1], [1, -1, -1], [1, -1, 1]], [[1, -1, -1], [1, 1, 1], [1, 1, -1]]]
    desc = {"width": cfg["width"], "height": cfg["height"], "spp": cfg["spp"], "max_bounce": cfg["max_bounce"],
            "integrator": cfg["integrator"], "camera": dict(zip(("rotatAngle", "upAngle", "r"), cfg["camera"])),
            "bvh": {"builder": "sah", "leaf": 8},
            "objects": [{"triangles": light, "material": {"baseColor": WHITE, "emissive": [12, 12, 12]}},
                        {"triangles": white, "material": {"baseColor": WHITE}},
                        {"triangles": red, "material": {"baseColor": RED}},
                        {"triangles": green, "material": {"baseColor": GREEN}}]}
    assert (desc["width"], desc["height"], desc["spp"]) == (256, 256, 1)
    (tmp_path / "c1.json").write_text(json.dumps(desc))
    built = render.build_scene(desc, str(tmp_path))
    canned = scenes.cornell_scene()
    assert built.tri.shape == (12, 36)
    assert np.array_equal(built.tri.view(np.uint32), canned.tri.view(np.uint32)) and np.array_equal(built.nodes, canned.nodes)
    assert render.main([str(tmp_path / "c1.json"), "-o", str(tmp_path / "c1.png"), "--pfm", str(tmp_path / "c1.pfm")]) == 0
    eye, cam = S.camera(*cfg["camera"])
    p = trace.make_params(256, 256, eye, cam, cfg["integrator"], cfg["max_bounce"], spp=1)
    want = canned.upload(oracle).render(p)
    got = imageio.read_pfm(tmp_path / "c1.pfm")
    assert np.array_equal(got.view(np.uint32), want[..., :3].view(np.uint32))
    png = _read_png_rgb8(tmp_path / "c1.png")
    want8 = oracle.tonemap(want.reshape(-1, 4)).reshape(256, 256, 3)[::-1]      # (PNG rows go top first)
    assert png.shape == (256, 256, 3) and np.array_equal(png, want8)
    # not a black frame: at ONE sample per pixel the light's own pixels (emission 12 -> saturated) and the few paths that found it
    lit = png.max(axis=2) > 0
    assert 0.01 < lit.mean() < 0.9 and int(png.max()) == 255 and len(np.unique(png.reshape(-1, 3), axis=0)) > 20
