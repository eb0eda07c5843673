"""tools/create_times.py [CFG]: ezrt_scene_create of a BASELINE scene, wall time per phase (EZRT_CREATE_TIMING=1) for EZRT_HOST_THREADS = 1 and the default."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, time; sys.path.insert(0, %r)
from ezrt_amd import scenes, trace
hip = trace.hip()
bs = {"C5": scenes.mega_scene, "C3": lambda: scenes.disney_grid_scene(subdiv=3), "C2": lambda: scenes.bunny_scene(subdiv=2)}[%r]()
for k in range(3):
    t = time.perf_counter(); sc = bs.upload(hip); d = time.perf_counter() - t
    print("ezrt_scene_create %%s: %%.3f s" %% (%r, d), file=sys.stderr); sc.close()
"""
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
for th in ("1", ""):
    env = dict(os.environ, EZRT_CREATE_TIMING="1")
    if th:
        env["EZRT_HOST_THREADS"] = th
    print("== host threads:", th or "default (<= 16)", flush=True)
    r = subprocess.run([sys.executable, "-c", code % (ROOT, cfg, cfg)], env=env, capture_output=True, text=True)
    print("\n".join(l for l in r.stderr.splitlines() if "ezrt" in l and "amdgpu" not in l)[-2500:], flush=True)
