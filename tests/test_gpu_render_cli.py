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
