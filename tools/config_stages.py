import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
name = sys.argv[1]
built = {"C3": lambda: scenes.disney_grid_scene(subdiv=3), "C4": lambda: scenes.p5_scene(subdiv=2), "C5": scenes.mega_scene}[name]()
cfg = scenes.CONFIGS[name]
sc = built.upload(hip)
eye, cam = S.camera(*cfg["camera"])
spp = int(sys.argv[2])
p = trace.make_params(cfg["width"], cfg["height"], eye, cam, cfg["integrator"], cfg["max_bounce"], spp=spp)
sc.render(p)
sc.render(p)
