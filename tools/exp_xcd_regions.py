#!/usr/bin/env python3
"""tools/exp_xcd_regions.py CFG [n_primary_millions]: would an XCD-AWARE bounce queue pay?

Synthetic bounce-1 rays of a BASELINE scene (as tools/exp_ray_order.py) traced by the TIMED kernel (ezrt_query_hits, audit_via_queue = 1) with the queue laid out so
that the workgroups of XCD x (workgroup b runs on XCD b mod 8) draw the rays whose ORIGINS lie in spatial region x -- eight equal-count regions along a Morton curve --
in random order inside the region (so waves stay mixed: sorting for coherence lost in round 4), against the same rays in a fully random order.  The trace queue is dealt
statically for this (EZRT_STATIC_PCT=95, EZRT_BOUNCE_SCATTER=0: position p -> pool p / 128 -> round k = pool / n_waves, wave (pool - 1223 k) mod n_waves -> workgroup
wave / 4 -> XCD).  Same rays, same answers (asserted)."""
import os
import sys

os.environ["EZRT_STATIC_PCT"] = "95"
os.environ["EZRT_BOUNCE_SCATTER"] = "0"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ezrt_amd import scene as S, scenes, trace  # noqa: E402
from exp_ray_order import camera_rays  # noqa: E402


def main():
    name = sys.argv[1].upper() if len(sys.argv) > 1 else "C3"
    mill = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    wg_per_cu = int(os.environ.get("WG_PER_CU", "6"))
    cfg = scenes.CONFIGS[name]
    bs = {"C2": lambda: scenes.bunny_scene(subdiv=2, hdr="shipped"), "C3": lambda: scenes.disney_grid_scene(subdiv=3, hdr="shipped"),
          "C4": lambda: scenes.p5_scene(subdiv=2, hdr="shipped"), "C5": lambda: scenes.mega_scene(hdr="shipped")}[name]()
    hip = trace.hip()
    sc = bs.upload(hip)
    sc.set_option("audit_via_queue", 1)
    eye, cam = S.camera(*cfg["camera"])
    W, H = cfg["width"], cfg["height"]
    frames = max(1, int(round(mill * 1e6 / (W * H))))
    rng = np.random.default_rng(7)
    prim = camera_rays(eye, cam, W, H, frames, rng)
    tri, t = sc.query_hits(prim)
    hit = tri >= 0
    d = prim[hit, 3:6]
    P = (prim[hit, 0:3] + d * t[hit, None]).astype(np.float32)
    T = bs.tri[tri[hit]]
    n = np.cross(T[:, 3:6] - T[:, 0:3], T[:, 6:9] - T[:, 0:3])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    n[(n * d).sum(1) > 0] *= -1.0
    u = rng.normal(size=P.shape)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    u[(u * n).sum(1) < 0] *= -1.0
    rays = np.concatenate([P, u.astype(np.float32)], 1).astype(np.float32)
    N = rays.shape[0]
    plo, phi = np.percentile(P, 1, axis=0), np.percentile(P, 99, axis=0)
    q = np.clip(((P - plo) / np.maximum(phi - plo, 1e-20) * 1024).astype(np.int64), 0, 1023)
    code = np.zeros(N, np.int64)
    for b in range(10):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1) << (3 * b + k)
    by_curve = np.argsort(code, kind="stable")
    region = np.empty(N, np.int64)
    region[by_curve] = np.arange(N) * 8 // N          # eight equal-count regions along the curve
    # which XCD draws queue position p (static rounds of the persistent kernel: ezrt_traceq4.h)
    n_waves = 256 * wg_per_cu * 4
    pool = 128
    pos = np.arange(N)
    pq = pos // pool
    k = pq // n_waves
    w = (pq % n_waves - 1223 * k) % n_waves
    xcd = (w // 4) % 8
    orders = {"random": rng.permutation(N)}
    for label, reg in (("regions = 8 Morton ranges of the ORIGIN, one per XCD, random inside", region),
                       ("control: 8 RANDOM groups, one per XCD (the layout alone)", rng.integers(0, 8, N))):
        o = np.empty(N, np.int64)
        spill = []
        for x in range(8):
            mine = rng.permutation(np.nonzero(reg == x)[0])
            slots = np.nonzero(xcd == x)[0]
            m = min(len(mine), len(slots))
            o[slots[:m]] = mine[:m]
            spill.append((slots[m:], mine[m:]))
        free = np.concatenate([s for s, _ in spill])
        left = rng.permutation(np.concatenate([r for _, r in spill]))
        o[free] = left
        orders[label] = o
        if reg is region:
            print("  (rays outside their XCD's slots: %.1f %%)" % (100.0 * len(left) / N))
    want = None
    for label, o in orders.items():
        r = np.ascontiguousarray(rays[o])
        best = 1e9
        for _ in range(3):
            tr, tt = sc.query_hits(r)
            _, ms, _ = sc.last_render_ms()
            best = min(best, ms)
        inv = np.empty(N, np.int64)
        inv[o] = np.arange(N)
        if want is None:
            want = (tr[inv].copy(), tt[inv].copy())
        else:
            assert np.array_equal(tr[inv], want[0]) and np.array_equal(tt[inv].view(np.uint32), want[1].view(np.uint32))
        print("  %-72s %8.3f ms  %7.2f Grays/s  (%d rays)" % (label, best, N / best / 1e6, N), flush=True)


if __name__ == "__main__":
    main()
