"""tools/slot_fill.py CFG: used slots per 4-wide record of a BASELINE scene (records4 from ezrt_scene_prune_info; every record but the
root and every reachable leaf is referenced by exactly one slot)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ezrt_amd import scenes, trace
hip = trace.hip()
name = sys.argv[1]
built = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
         "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
sc = built.upload(hip)
info = sc.prune_info()
nodes = built.nodes
n = nodes[1:, 3]  # leaf triangle counts (0 for inner nodes)
leaves = int((n > 0).sum())
rec = int(info["records4"])
print(name, "records4", rec, "leaves", leaves, "inner(binary)", int((n == 0).sum()), "slots used per record %.3f" % ((rec - 1 + leaves) / rec),
      "triangles per leaf %.2f" % (built.tri.shape[0] / leaves))
