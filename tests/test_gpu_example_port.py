"""examples/p5_main_port.cpp = chapter 5's main() + display() (P5/main.cpp:697-748, 763-947) ported onto the C ABI
(include/ezrt.h) and the C++ host scene API (include/ezrt_scene.hpp): a compiled consumer of the boundary that is not
Python.  Its frame must be the one the ctypes binding renders from the same files."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "examples", "bin", "p5_main_port")


def _build():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", ROOT, "examples"])


@pytest.mark.gpu
def test_p5_main_port_renders_the_frame_of_the_python_binding(hip, tmp_path):
    from ezrt_amd import imageio, scene as S, scenes, trace
    from ezrt_amd._abi import INTEGRATOR_P5_MIS
    _build()
    bv, bf = scenes.mesh("bunny")
    qv, qf = scenes.mesh("quad")
    (tmp_path / "bunny.obj").write_bytes(scenes.obj_text(bv, bf))
    (tmp_path / "quad.obj").write_bytes(scenes.obj_text(qv, qf))
    with np.load(scenes.SHIPPED_HDR) as z:
        imageio.write_hdr_rgbe(str(tmp_path / "env.hdr"), z["rgbe"])
    size, frames = 128, 12
    out = subprocess.run([BIN, str(tmp_path / "bunny.obj"), str(tmp_path / "quad.obj"), str(tmp_path / "env.hdr"),
                          str(tmp_path / "out.pfm"), str(size), str(frames), str(tmp_path / "out.ppm"),
                          "0.1", "-1.0", "0", "0.75"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "backend hip:gfx950" in out.stdout
    got = imageio.read_pfm(str(tmp_path / "out.pfm"))

    # the same scene through the Python binding (scenes.p5_scene = the same readObj/buildBVHwithSAH sequence)
    bs = scenes.p5_scene(subdiv=0, hdr="shipped")
    sg = bs.upload(hip)
    eye, cam = S.camera(90, 10, 2.0)
    p = trace.make_params(size, size, eye, cam, INTEGRATOR_P5_MIS, 2, spp=frames)
    want = sg.render(p)
    assert got.shape == (size, size, 3)
    assert np.array_equal(got, want[..., :3]), float(np.max(np.abs(got - want[..., :3])))
    assert np.isfinite(got).all() and float(got.max()) > 0.1

    # pass3 + 8-bit quantisation written as a PPM, top row first
    ppm = (tmp_path / "out.ppm").read_bytes()
    head = b"P6\n%d %d\n255\n" % (size, size)
    assert ppm.startswith(head)
    rgb8 = np.frombuffer(ppm[len(head):], np.uint8).reshape(size, size, 3)
    assert np.array_equal(rgb8[::-1], hip.tonemap(want.reshape(-1, 4)).reshape(size, size, 3))


def test_c_translation_unit_pins_the_struct_layout():
    """examples/abi_layout_check.c: C11 _Static_asserts on EzrtRenderParams (136 bytes, field offsets) and the record
    sizes, linked against both libraries (no GPU call)."""
    _build()
    out = subprocess.run([os.path.join(ROOT, "examples", "bin", "abi_layout_check")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sizeof(EzrtRenderParams)=136" in out.stdout
