"""tools/debug_stages.py [R N]: per-stage queue sizes (EZRT_DEBUG_STAGES=1) / per-wave life times, iterations, refills, steals of every trace
launch (=2) of one C2 call (64 spp), optionally of shard R of N only.  INSTR=1 runs the instrumented (binary, unpruned) route."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=2, hdr="shipped")
sc = bs.upload(hip)
eye, cam = S.camera(0, 0, 4)
sc.set_instrumentation(int(os.environ.get("INSTR", "0")))
shard = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
p = trace.make_params(512, 512, eye, cam, 50, 4, spp=64, tile=(16, 16), shard=shard)
sc.render(p)
