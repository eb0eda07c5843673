"""numpy ORACLE for the host-side algorithms and the intersector.  TEST INFRASTRUCTURE ONLY.

Independent float32 restatements, used by tests/ to pin
  * the SAH / median BVH builders   (P3/main.cpp:394-588)
  * calculateHdrCache                (P5/main.cpp:592-689)
  * hitTriangle / brute-force scan   (P2/main.cpp:212-238, 436-446; "暴力验证" P2/main.cpp:585)
against the C++ host library and the C oracle.  Pure-Python loops: small cases only.
PARITY PINNING: the reference has no tests for these; std::sort's order of equal keys is
implementation-defined, so builder comparisons are only made on inputs with distinct
centroid keys (asserted by the caller).
"""
import numpy as np

f32 = np.float32
INF = f32(114514.0)


# ----------------------------------------------------------------------------- builders
def _gmin(a, b):
    return b if b < a else a


def _gmax(a, b):
    return b if a < b else a


def _tri_bounds(P):
    """P: [n, 3 verts, 3] float32 -> lo, hi [n, 3] with glm min/max nesting min(p1, min(p2, p3))."""
    lo = np.where(np.minimum(P[:, 1], P[:, 2]) < P[:, 0], np.minimum(P[:, 1], P[:, 2]), P[:, 0])
    hi = np.where(P[:, 0] < np.maximum(P[:, 1], P[:, 2]), np.maximum(P[:, 1], P[:, 2]), P[:, 0])
    return lo.astype(f32), hi.astype(f32)


def _centroids(P):
    return (((P[:, 0] + P[:, 1]) + P[:, 2]) / f32(3.0)).astype(f32)


def build_bvh(tri36, leaf_n=8, sah=True):
    """Returns (order, nodes): order = permutation of the input triangles, nodes = list of
    [left, right, n, index, AA(3), BB(3)] with the testNode dummy at index 0."""
    P = np.asarray(tri36, f32).reshape(-1, 36)[:, :9].reshape(-1, 3, 3)
    n = P.shape[0]
    lo, hi = _tri_bounds(P)
    cen = _centroids(P)
    order = list(range(n))
    nodes = [[255, 128, 30, 0, (1, 1, 0), (0, 1, 0)]]

    def sort_axis(l, r, axis):
        seg = order[l:r + 1]
        keys = [cen[s][axis] for s in seg]
        assert len(set(float(k) for k in keys)) == len(keys), "equal centroid keys: order is STL-defined"
        seg.sort(key=lambda s: cen[s][axis])
        order[l:r + 1] = seg

    def new_node(l, r):
        AA = [f32(1145141919)] * 3
        BB = [f32(-1145141919)] * 3
        for i in range(l, r + 1):
            s = order[i]
            for k in range(3):
                AA[k] = _gmin(AA[k], lo[s][k])
                BB[k] = _gmax(BB[k], hi[s][k])
        nodes.append([0, 0, 0, 0, tuple(AA), tuple(BB)])
        return len(nodes) - 1

    def area_cost(mn, mx, cnt):
        lx, ly, lz = mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]
        S = f32(2.0) * ((lx * ly) + (lx * lz) + (ly * lz))
        return f32(S * f32(cnt))

    def rec(l, r):
        if l > r:
            return 0
        nid = new_node(l, r)
        if r - l + 1 <= leaf_n:
            nodes[nid][2] = r - l + 1
            nodes[nid][3] = l
            return nid
        if not sah:
            AA, BB = nodes[nid][4], nodes[nid][5]
            lx, ly, lz = BB[0] - AA[0], BB[1] - AA[1], BB[2] - AA[2]
            if lx >= ly and lx >= lz:
                sort_axis(l, r, 0)
            if ly >= lx and ly >= lz:
                sort_axis(l, r, 1)
            if lz >= lx and lz >= ly:
                sort_axis(l, r, 2)
            split = (l + r) // 2
        else:
            Cost, Axis, Split = INF, 0, (l + r) // 2
            for axis in range(3):
                sort_axis(l, r, axis)
                cnt = r - l + 1
                lmx, lmn, rmx, rmn = [None] * cnt, [None] * cnt, [None] * cnt, [None] * cnt
                mx, mn = [-INF] * 3, [INF] * 3
                for i in range(l, r + 1):
                    s = order[i]
                    mx = [_gmax(mx[k], hi[s][k]) for k in range(3)]
                    mn = [_gmin(mn[k], lo[s][k]) for k in range(3)]
                    lmx[i - l], lmn[i - l] = mx, mn
                mx, mn = [-INF] * 3, [INF] * 3
                for i in range(r, l - 1, -1):
                    s = order[i]
                    mx = [_gmax(mx[k], hi[s][k]) for k in range(3)]
                    mn = [_gmin(mn[k], lo[s][k]) for k in range(3)]
                    rmx[i - l], rmn[i - l] = mx, mn
                cost, split = INF, l
                for i in range(l, r):
                    total = f32(area_cost(lmn[i - l], lmx[i - l], i - l + 1) + area_cost(rmn[i + 1 - l], rmx[i + 1 - l], r - i))
                    if total < cost:
                        cost, split = total, i
                if cost < Cost:
                    Cost, Axis, Split = cost, axis, split
            sort_axis(l, r, Axis)
            split = Split
        left = rec(l, split)
        right = rec(split + 1, r)
        nodes[nid][0], nodes[nid][1] = left, right
        return nid

    rec(0, n - 1)
    return order, nodes


def encode_nodes(nodes):
    out = np.zeros((len(nodes), 12), f32)
    for i, (l, r, n, idx, AA, BB) in enumerate(nodes):
        out[i] = [l, r, 0, n, idx, 0, AA[0], AA[1], AA[2], BB[0], BB[1], BB[2]]
    return out


# ----------------------------------------------------------------------------- hdr cache
def hdr_cache(hdr):
    """P5/main.cpp:592-689 with fp32 running sums in the reference's loop order."""
    hdr = np.asarray(hdr, f32)
    H, W, _ = hdr.shape
    lum = (0.2 * hdr[..., 0].astype(np.float64) + 0.7 * hdr[..., 1].astype(np.float64)
           + 0.1 * hdr[..., 2].astype(np.float64)).astype(f32)
    lum_sum = f32(0)
    for i in range(H):
        for j in range(W):
            lum_sum = f32(lum_sum + lum[i, j])
    pdf = (lum / lum_sum).astype(f32)
    margin = np.zeros(W, f32)
    for i in range(H):
        margin = (margin + pdf[i]).astype(f32)
    cdf_x = np.cumsum(margin, dtype=f32)  # sequential fp32 adds
    cond = (pdf / margin[None, :]).astype(f32)
    cdf_y = np.cumsum(cond, axis=0, dtype=f32)
    out = np.zeros((H, W, 3), f32)
    for i in range(H):
        xi1 = f32(i) / f32(H)
        x = int(np.searchsorted(cdf_x, xi1, side="left"))
        x = min(x, W - 1)
        for j in range(W):
            xi2 = f32(j) / f32(W)
            y = int(np.searchsorted(cdf_y[:, x], xi2, side="left"))
            out[i, j] = (f32(x) / f32(W), f32(y) / f32(H), pdf[i, j])
    return out


# ----------------------------------------------------------------------------- intersector
def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(f32)


def hit_triangles(tri36, S, d):
    """hitTriangle (P2/main.cpp:212-238) for one ray against all triangles -> t [n] (INF = miss)."""
    T = np.asarray(tri36, f32).reshape(-1, 36)
    p1, p2, p3 = T[:, 0:3], T[:, 3:6], T[:, 6:9]
    S = np.asarray(S, f32)[None, :]
    d = np.asarray(d, f32)[None, :]
    with np.errstate(all="ignore"):
        c = _cross(p2 - p1, p3 - p1)
        inv = (f32(1.0) / np.sqrt(_dot(c, c).astype(f32))).astype(f32)
        N = (c * inv[:, None]).astype(f32)
        flip = _dot(N, d) > 0
        N = np.where(flip[:, None], -N, N)
        Nd = _dot(N, d)
        t = ((_dot(N, p1) - _dot(S, N)) / _dot(d, N)).astype(f32)
        P = (S + d * t[:, None]).astype(f32)
        s1 = _dot(_cross(p2 - p1, P - p1), N)
        s2 = _dot(_cross(p3 - p2, P - p2), N)
        s3 = _dot(_cross(p1 - p3, P - p3), N)
        inside = ((s1 > 0) & (s2 > 0) & (s3 > 0)) | ((s1 < 0) & (s2 < 0) & (s3 < 0))
        ok = (np.abs(Nd) >= f32(0.00001)) & (t >= f32(0.0005)) & inside
    return np.where(ok, t, INF).astype(f32)


def brute_force(tri36, S, d):
    """hitTriangleArray over everything: first index with the minimum t (strict <)."""
    t = hit_triangles(tri36, S, d)
    i = int(np.argmin(t))  # argmin returns the first minimum
    if not t[i] < INF:
        return -1, INF
    return i, t[i]
