import sys; sys.path.insert(0,'.')
import numpy as np
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=2)
sc = bs.upload(hip)
eye, cam = S.camera(0,0,4)
import os
sc.set_instrumentation(int(os.environ.get("INSTR","1")))
p = trace.make_params(512,512,eye,cam,50,4,spp=64)
sc.render(p)
