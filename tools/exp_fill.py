import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ezrt_amd import scenes, trace
hip = trace.hip()
for name, mk in (("C2", lambda: scenes.bunny_scene(subdiv=2, hdr="synthetic")), ("C3", lambda: scenes.disney_grid_scene(subdiv=3, hdr="synthetic")), ("C5", lambda: scenes.mega_scene(hdr="synthetic"))):
    b = mk()
    for r in ("0", "1"):
        os.environ["EZRT_RETREE"] = r
        sc = b.upload(hip)
        st, pi = sc.stats(), sc.prune_info()
        R, L = pi["records4"], st["n_leaves"]
        print(name, "retree", r, "leaves", L, "records", int(R), "avg slots used %.2f" % ((R - 1 + L) / R), "depth(binary ref)", st["depth"])
        sc.close()
