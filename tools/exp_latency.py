import sys, time; sys.path.insert(0,'.')
import numpy as np
from ezrt_amd import scene as S, scenes, trace
hip = trace.hip()
bs = scenes.bunny_scene(subdiv=2)
sc = bs.upload(hip)
P = bs.tri[:79488,:9].reshape(-1,3)
lo, hi = P.min(0), P.max(0)
rng = np.random.default_rng(0)
def rays(n):
    o = rng.uniform(lo, hi, (n,3)).astype(np.float32)
    d = rng.normal(size=(n,3)).astype(np.float32); d /= np.linalg.norm(d,axis=1,keepdims=True)
    return np.concatenate([o,d.astype(np.float32)],1)
sc.set_instrumentation(1)
for n in (1, 64, 256, 4096, 65536, 1<<20):
    r = rays(n)
    sc.query_hits(r[:1])
    best = 1e9
    for _ in range(5):
        sc.counters_reset()
        t=time.perf_counter(); sc.query_hits(r); dt=time.perf_counter()-t
        best=min(best,dt)
    c = sc.counters()
    print("n=%8d  best %.1f us   pops/ray %.1f tris/ray %.1f" % (n, best*1e6, c['node_pops']/n, c['tri_tests']/n))
