"""Differential fuzz of the nearest-hit query (rr_debug_trace, device BVH4 traversal, both builders)
against the oracle's brute-force loop: bit-exact (t, face) expected.  Nasty inputs on purpose: triangle
sizes from 1e-3 to 100 m in one scene, coplanar quads sharing edges, duplicated triangles (tie -> lower face
id), axis-parallel rays, rays aimed exactly at vertices / edge midpoints, origins on triangle planes.
Seeds >= 100000: a terrain crossed by huge thin faces (the host builder then files faces under spatially split, clipped boxes) and
rays that graze those faces next to their edges -- the residual class of the grazing guard.
usage: fuzz_trace.py [n_seeds] [first_seed]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native
from oracle import oracle


def scene(rs):
    kind = rs.randint(0, 4)
    if kind == 0:      # soup with wildly different sizes
        n = int(rs.randint(50, 1500))
        cen = rs.uniform(-40, 40, (n, 1, 3))
        size = 10.0 ** rs.uniform(-3, 2, (n, 1, 1))
        v = (cen + rs.normal(0, 1, (n, 3, 3)) * size).astype(np.float32)
    elif kind == 1:    # planar grid of quads (shared edges and vertices), a few tilted copies
        m = int(rs.randint(3, 25)); g = np.linspace(-20, 20, m + 1)
        tris = []
        for z, tilt in ((0.0, 0.0), (5.0, 0.1), (-3.0, -0.2)):
            for i in range(m):
                for j in range(m):
                    p = [[g[i], g[j], z + tilt * g[i]], [g[i + 1], g[j], z + tilt * g[i + 1]],
                         [g[i + 1], g[j + 1], z + tilt * g[i + 1]], [g[i], g[j + 1], z + tilt * g[i]]]
                    tris += [[p[0], p[1], p[2]], [p[0], p[2], p[3]]]
        v = np.array(tris, np.float32)
    elif kind == 2:    # duplicates and near-duplicates
        n = int(rs.randint(20, 300))
        base = (rs.uniform(-15, 15, (n, 1, 3)) + rs.normal(0, 3, (n, 3, 3))).astype(np.float32)
        v = np.concatenate([base, base, base + np.float32(1e-6), base[::-1]])
    else:              # closed boxes
        tris = []
        for _ in range(int(rs.randint(1, 40))):
            lo = rs.uniform(-30, 25, 3); hi = lo + rs.uniform(0.01, 8, 3)
            c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                          [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
            for a, b, cc, d in ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 3, 7, 4)):
                tris += [[c[a], c[b], c[cc]], [c[a], c[cc], c[d]]]
        v = np.array(tris, np.float32)
    n = len(v)
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def split_scene(rs):
    """Seeds >= 100000 (round 6; advisor, round 5): a fine terrain crossed by a few huge, thin faces -- what makes the host
    builder CUT faces at planes (spatial splits: a cut face is filed under clipped boxes) -- for rays that graze those faces."""
    m = int(rs.randint(12, 40)); g = np.linspace(-20, 20, m + 1)
    z = rs.normal(0, 0.15, (m + 1, m + 1))
    tris = []
    for i in range(m):
        for j in range(m):
            p = [[g[i], g[j], z[i, j]], [g[i + 1], g[j], z[i + 1, j]], [g[i + 1], g[j + 1], z[i + 1, j + 1]], [g[i], g[j + 1], z[i, j + 1]]]
            tris += [[p[0], p[1], p[2]], [p[0], p[2], p[3]]]
    for _ in range(int(rs.randint(4, 16))):           # long slivers and big walls through the terrain
        a = rs.uniform(-19, 19, 3); a[2] = rs.uniform(-0.5, 0.5)
        dirn = rs.normal(0, 1, 3); dirn[2] *= 0.05; dirn /= np.linalg.norm(dirn)
        side = np.cross(dirn, [0, 0, 1.0]) * rs.uniform(0.01, 6.0) + np.array([0, 0, rs.uniform(0.0, 3.0)])
        L = rs.uniform(10, 38)
        q = [a - 0.5 * L * dirn, a + 0.5 * L * dirn, a + 0.5 * L * dirn + side, a - 0.5 * L * dirn + side]
        tris += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
    v = np.array(tris, np.float32)
    n = len(v)
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def grazing_rays(rs, v, f, n=3000):
    """rays within 1e-6 .. 6e-3 rad of the plane of one of the LARGEST faces, passing within a few centimetres of its edges"""
    tri = v[f].astype(np.float64)
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    big = np.argsort(area)[-max(4, len(tri) // 200):]
    pick = big[rs.randint(0, len(big), n)]
    t0, e1, e2 = tri[pick, 0], tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0]
    nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # a point on an edge (or just beside it), the ray runs in the plane through it
    u = rs.uniform(0, 1, (n, 1)); w = rs.choice([0.0, 1.0], (n, 1))
    edge_pt = t0 + np.where(w == 0, u * e1, u * e2) + rs.normal(0, 0.02, (n, 3))
    inplane = rs.uniform(-1, 1, (n, 1)) * e1 + rs.uniform(-1, 1, (n, 1)) * e2
    inplane /= np.linalg.norm(inplane, axis=1, keepdims=True)
    tilt = 10.0 ** rs.uniform(-6, -2.2, (n, 1)) * rs.choice([-1.0, 1.0], (n, 1))
    d = inplane + tilt * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = edge_pt - d * rs.uniform(1.0, 30.0, (n, 1))
    return o.astype(np.float32), d.astype(np.float32)


def rays(rs, v, f, n=3000):
    tri = v[f]                                   # [n][3][3]
    o = rs.uniform(-45, 45, (n, 3)).astype(np.float32)
    d = rs.normal(0, 1, (n, 3))
    k = n // 6
    pick = rs.randint(0, len(tri), n)
    # aimed exactly at a vertex / an edge midpoint / the centroid of a random triangle
    tgt = np.where(rs.randint(0, 3, (n, 1)) == 0, tri[pick, 0], np.where(rs.randint(0, 2, (n, 1)) == 0,
                   0.5 * (tri[pick, 0] + tri[pick, 1]), tri[pick].mean(axis=1)))
    d[:3 * k] = (tgt - o)[:3 * k]
    # axis-parallel (two zero components) and plane-parallel (one zero component)
    d[3 * k:4 * k] = np.eye(3)[rs.randint(0, 3, k)] * rs.choice([-1.0, 1.0], (k, 1))
    d[4 * k:5 * k, rs.randint(0, 3)] = 0.0
    # origins ON a triangle's plane (at its centroid), direction random
    o[5 * k:] = tri[pick[5 * k:]].mean(axis=1)
    nrm = np.linalg.norm(d, axis=1, keepdims=True); nrm[nrm == 0] = 1
    return o.astype(np.float32), (d / nrm).astype(np.float32)


def run(n_seeds=20, first=0, verbose=True):
    bad = 0
    for seed in range(first, first + n_seeds):
        rs = np.random.RandomState(seed)
        if seed >= 100000:               # spatial splits + grazing incidence (split_scene)
            v, f = split_scene(rs)
            o1, d1 = grazing_rays(rs, v, f, 2000); o2, d2 = rays(rs, v, f, 1000)
            o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
        else:
            v, f = scene(rs)
            o, d = rays(rs, v, f)
        brute = oracle.Scene(v, f, None, use_bvh=0)
        want_t = np.full(len(o), -1.0, np.float32); want_f = np.full(len(o), 0xFFFFFFFF, np.uint32)
        for i in range(len(o)):
            r = brute.intersect(o[i], d[i])
            if r is not None:
                want_t[i], want_f[i] = r[0], r[1]
        for builder in ("host", "gpu"):
            c = native.Context(0)
            c.set_mesh(v, f, None, builder=builder)
            t, face = c.debug_trace(o, d)
            c.close()
            hit = want_t >= 0
            ok = np.array_equal(t[hit], want_t[hit]) and np.array_equal(face[hit], want_f[hit]) and (t[~hit] < 0).all()
            if not ok:
                # (Until round 4 one class of difference was reported apart: a ray that grazes a triangle's plane makes
                # Moeller-Trumbore accept a point outside the triangle's padded box, which a hierarchy skips -- seed 307.  The
                # grazing guard of round 5 made the hit definition independent of the structure: every difference is a mismatch.)
                idx = np.nonzero((t != want_t) | ((face != want_f) & hit))[0]
                bad += 1
                i = int(idx[0])
                print("MISMATCH seed", seed, builder, "ray", i, "gpu", t[i], face[i], "oracle", want_t[i], want_f[i], "o", o[i], "d", d[i])
    if verbose:
        print("trace fuzz: %d scenes x 2 builders, %d mismatching" % (n_seeds, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
