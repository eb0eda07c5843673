#!/usr/bin/env python3
"""tools/update_design_results.py [rN]: replace DESIGN.md section 6's "Results on MI355X, round N" paragraph by the one tools/results_paragraph.py derives from profiles/rN/
(keeping the sentences that follow the derived text: the GPU suite / routes / gloo lines)."""
import os, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r6"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "results_paragraph.py"), R], text=True)
para = out[out.index("Results on MI355X, round"):].rstrip("\n")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
i = s.index("Results on MI355X, round %s (" % R[1:])
j = s.index("The GPU suite on that box:", i)
s = s[:i] + para + "  " + s[j:]
open(p, "w").write(s)
print("updated")
