"""Progressive accumulation with checkpoint / resume: what display() and the mouse
handlers of the reference do for a user (P3/main.cpp:596-672), minus the window.

  display()            one more sample per pixel into lastFrame, frameCounter++     -> step(n)
  mouse(x, y)          frameCounter = 0; rotatAngle / upAngle from the drag, upAngle clamped to
                       [-89, 89] (P3/main.cpp:650-660)                               -> drag(dx, dy)
  mouseWheel(dir)      frameCounter = 0; r += -direction * 0.5 (P3/main.cpp:670-674) -> wheel(direction)

A camera change only resets the counter: frame 0 ignores the previous frame-buffer content (the
reference multiplies it by 0), so the buffer is not cleared.  A checkpoint is {frame buffer, frame
counter, camera, render settings}; resuming from it continues the running mean exactly where it
stopped -- the result is bit-identical to an uninterrupted run (tests/test_progressive.py).

lastFrame lives on the device (ezrt_frame_create, like the GL texture it replaces): step() only enqueues
ezrt_render_device, and the frame crosses PCIe when `accum` is read (present / save), not once per display() call."""
import json

import numpy as np

from . import scene as S
from . import trace


class ProgressiveRenderer:
    def __init__(self, gpu_scene, width=512, height=512, integrator=50, max_bounce=4, env_clamp=0.0,
                 rotatAngle=0.0, upAngle=0.0, r=4.0):
        self.scene = gpu_scene
        self.width, self.height = int(width), int(height)
        self.integrator, self.max_bounce = int(integrator), int(max_bounce)
        self.env_clamp = env_clamp
        self.rotatAngle, self.upAngle, self.r = float(rotatAngle), float(upAngle), float(r)
        self.frameCounter = 0
        self.frame = gpu_scene._tl.frame(self.width, self.height)   # lastFrame, device-resident

    @property
    def accum(self):
        """lastFrame as a host array (a synchronising read-back)."""
        return self.frame.read()

    # ---- the reference's callbacks
    def step(self, n=1):
        """n more samples per pixel (n calls of display()); asynchronous, lastFrame stays on the device."""
        eye, cam = S.camera(self.rotatAngle, self.upAngle, self.r)
        p = trace.make_params(self.width, self.height, eye, cam, self.integrator, self.max_bounce, spp=int(n),
                              frame0=self.frameCounter, env_clamp=self.env_clamp)
        self.scene.render_device(p, self.frame.ptr)
        self.frameCounter += int(n)

    def drag(self, dx, dy):
        self.frameCounter = 0
        self.rotatAngle += 150.0 * dx / 512.0
        self.upAngle += 150.0 * dy / 512.0
        self.upAngle = max(min(self.upAngle, 89.0), -89.0)

    def wheel(self, direction):
        self.frameCounter = 0
        self.r += -direction * 0.5

    # ---- checkpoint / resume
    def settings(self):
        return {"width": self.width, "height": self.height, "integrator": self.integrator,
                "max_bounce": self.max_bounce, "env_clamp": self.env_clamp, "rotatAngle": self.rotatAngle,
                "upAngle": self.upAngle, "r": self.r, "frameCounter": self.frameCounter}

    def save(self, path):
        np.savez(path, accum=self.accum, settings=np.frombuffer(json.dumps(self.settings()).encode(), np.uint8))

    @classmethod
    def load(cls, path, gpu_scene):
        with np.load(path) as z:
            st = json.loads(bytes(z["settings"]).decode())
            accum = np.ascontiguousarray(z["accum"], np.float32)
        fc = st.pop("frameCounter")
        self = cls(gpu_scene, **st)
        if accum.shape != (self.height, self.width, 4):
            raise ValueError("checkpoint frame buffer %s does not match %dx%d" % (accum.shape, self.width, self.height))
        self.frame.write(accum)
        self.frameCounter = int(fc)
        return self
