#!/usr/bin/env python3
"""tools/exp_ray_order.py CFG [n_primary_millions]: what would ray-coherence binning of a bounce stage buy?

Synthetic bounce-1 rays of a BASELINE scene (primary hit points of its camera, uniform-hemisphere directions about the geometric
normal) are traced by the TIMED kernel (traceq4_kernel through ezrt_query_hits, audit_via_queue = 1) in different queue ORDERS;
the trace launch is timed with stream events (ezrt_last_render_ms).  Orders: pipeline (8x8 sub-blocks of a frame, scattered),
random, cell-sorted (4^3 .. 16^3 cells of the scene box), cell + octant, octant only, Morton, and batches of 512 sorted locally."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ezrt_amd import scene as S, scenes, trace  # noqa: E402


def camera_rays(eye, cam, w, h, frames, rng):
    out = []
    m = np.asarray(cam, np.float64).reshape(4, 4).T
    for _ in range(frames):
        xs, ys = np.meshgrid(np.arange(w), np.arange(h))
        # 8x8 sub-block order inside the frame (what a queue granule of the pipeline holds)
        bx, by = xs // 8, ys // 8
        key = (by * (w // 8) + bx) * 64 + (ys % 8) * 8 + (xs % 8)
        order = np.argsort(key.ravel(), kind="stable")
        px = ((xs + 0.5) / w * 2 - 1 + rng.uniform(-0.5, 0.5, xs.shape) / w).ravel()[order]
        py = ((ys + 0.5) / h * 2 - 1 + rng.uniform(-0.5, 0.5, xs.shape) / h).ravel()[order]
        d = px[:, None] * m[:3, 0] + py[:, None] * m[:3, 1] - 1.5 * m[:3, 2]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        out.append(d.astype(np.float32))
    d = np.concatenate(out)
    o = np.broadcast_to(np.asarray(eye, np.float32), d.shape)
    return np.concatenate([o, d], 1).astype(np.float32)


def main():
    name = sys.argv[1].upper() if len(sys.argv) > 1 else "C2"
    mill = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
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
    _, tr_ms, _ = sc.last_render_ms()
    print("%s: %d primary rays (%d frames), %.1f %% hit; primary trace launch %.3f ms = %.2f Grays/s" %
          (name, prim.shape[0], frames, 100.0 * (tri >= 0).mean(), tr_ms, prim.shape[0] / tr_ms / 1e6), flush=True)
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
    lo, hi = bs.tri[:, :9].reshape(-1, 3).min(0), bs.tri[:, :9].reshape(-1, 3).max(0)
    # (the scene box is dominated by the floor: cells over the box of the hit points' 1..99 percentiles, clamped)
    plo, phi = np.percentile(P, 1, axis=0), np.percentile(P, 99, axis=0)

    def cell(res, a=plo, b=phi):
        q = np.clip(((P - a) / np.maximum(b - a, 1e-20) * res).astype(np.int64), 0, res - 1)
        return (q[:, 0] * res + q[:, 1]) * res + q[:, 2]

    def morton(bits=10):
        q = np.clip(((P - plo) / np.maximum(phi - plo, 1e-20) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
        code = np.zeros(N, np.int64)
        for b in range(bits):
            for k in range(3):
                code |= ((q[:, k] >> b) & 1) << (3 * b + k)
        return code

    octant = ((u[:, 0] < 0).astype(np.int64) | ((u[:, 1] < 0).astype(np.int64) << 1) | ((u[:, 2] < 0).astype(np.int64) << 2))
    # pipeline order: granules of 64 primary slots (one 8x8 sub-block of one frame) scattered
    slot = np.nonzero(hit)[0]
    gran = slot // 64
    perm_g = rng.permutation(prim.shape[0] // 64 + 1)
    pipeline = np.argsort(perm_g[gran], kind="stable")
    orders = {"pipeline (scattered 8x8 sub-blocks)": pipeline, "random": rng.permutation(N)}
    base = pipeline
    for res in (4, 8, 16, 32):
        orders["cell %d^3" % res] = base[np.argsort(cell(res)[base], kind="stable")]
    orders["cell 16^3 + octant"] = base[np.argsort((cell(16) * 8 + octant)[base], kind="stable")]
    orders["cell 8^3 + octant"] = base[np.argsort((cell(8) * 8 + octant)[base], kind="stable")]
    orders["octant + cell 8^3"] = base[np.argsort((octant * 512 + cell(8))[base], kind="stable")]
    orders["octant only"] = base[np.argsort(octant[base], kind="stable")]
    orders["morton 30 bit"] = base[np.argsort(morton()[base], kind="stable")]
    orders["morton 30 bit + octant (low bits)"] = base[np.argsort((morton() >> 9) * 8 + octant, kind="stable")] if False else base[np.argsort(((morton() >> 9) * 8 + octant)[base], kind="stable")]
    for bsz in (512, 2048):
        k = (cell(16) * 8 + octant)[base]
        blk = np.arange(N) // bsz
        orders["pipeline, batches of %d sorted by cell 16^3 + octant" % bsz] = base[np.lexsort((k, blk))]
    if os.environ.get("EXP_ORDERS", "") == "granules":   # second round: how fine must the mixing be?  (sorting lost: profiles/r4)
        orders = {"pipeline (scattered 8x8 sub-blocks)": pipeline, "random": rng.permutation(N)}
        for g in (8, 16, 32, 64, 128):
            ng = (N + g - 1) // g
            pg = rng.permutation(ng)
            idx = (pg[:, None] * g + np.arange(g)[None, :]).ravel()
            orders["pipeline, granules of %d rays shuffled" % g] = base[idx[idx < N]]
        for g, M in ((16, 251), (16, 2531), (32, 251), (8, 251)):
            ng = (N + g - 1) // g
            while ng % M == 0:
                M += 2
            lg = np.arange(ng)
            pg = (lg * M) % ng
            idx = (pg[:, None] * g + np.arange(g)[None, :]).ravel()
            orders["pipeline, granule of %d -> (granule x %d) mod n" % (g, M)] = base[idx[idx < N]]
    want = None
    for label, o in orders.items():
        r = np.ascontiguousarray(rays[o])
        best = 1e9
        for _ in range(3):
            tr, tt = sc.query_hits(r)
            _, ms, _ = sc.last_render_ms()
            best = min(best, ms)
        if want is None:
            want = (tr.copy(), tt.copy(), o)
        else:   # same rays, another order: same answers
            inv = np.empty(N, np.int64)
            inv[o] = np.arange(N)
            w_inv = np.empty(N, np.int64)
            w_inv[want[2]] = np.arange(N)
            assert np.array_equal(tr[inv], want[0][w_inv]) and np.array_equal(tt[inv].view(np.uint32), want[1][w_inv].view(np.uint32))
        print("  %-62s %8.3f ms  %7.2f Grays/s  (%d rays, %.1f %% hit)" % (label, best, N / best / 1e6, N, 100.0 * (tr >= 0).mean()), flush=True)


if __name__ == "__main__":
    main()
