"""fmhr_amd.subdivide — Loop subdivision with the index contract the reference relies on (the reference
calls trimesh.remesh.subdivide_loop(iterations=3) at /root/reference/mesh_sfs_optim.py:82,106 and repose.py:14-24 depends on the
child-face layout).  Host-side numpy: topology preprocessing runs once per capture.

Contract: original ("even") vertices keep indices 0..V-1; one new ("odd") vertex per unique edge is appended in unique-edge
order (sorted by (min, max) vertex index); each parent [f0,f1,f2] becomes the four CONSECUTIVE children
[f0,m01,m20], [m01,f1,m12], [m20,m12,f2], [m01,m12,m20].
Positions: odd = 3/8 (a+b) + 1/8 (c+d) on interior edges, 1/2 (a+b) on boundary edges; even = (1-k*beta) v + beta * sum(neighbours)
with beta = (40 - (3 + 2 cos(2 pi / k))^2) / (64 k) for interior valence k, 1/8 a + 3/4 v + 1/8 b on the boundary."""
import numpy as np


def _edges(faces, V):
    f = faces.astype(np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    opp = np.concatenate([f[:, 2], f[:, 0], f[:, 1]])
    lo, hi = e.min(1), e.max(1)
    key = lo * V + hi
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    return uniq, inv, cnt, opp, np.stack([uniq // V, uniq % V], 1)


def subdivide_loop_once(verts, faces):
    V, F = verts.shape[0], faces.shape[0]
    uniq, inv, cnt, opp, edges = _edges(faces, V)
    E = uniq.shape[0]
    a, b = edges[:, 0], edges[:, 1]
    interior = cnt == 2
    # odd vertices
    opp_sum = np.zeros((E, verts.shape[1]))
    np.add.at(opp_sum, inv, verts[opp])
    odd = 0.5 * (verts[a] + verts[b])
    odd[interior] = 0.375 * (verts[a[interior]] + verts[b[interior]]) + 0.125 * opp_sum[interior]
    # even vertices
    nb_sum = np.zeros_like(verts, dtype=np.float64)
    val = np.zeros(V)
    np.add.at(nb_sum, a, verts[b]); np.add.at(nb_sum, b, verts[a])
    np.add.at(val, a, 1); np.add.at(val, b, 1)
    bd = ~interior
    on_boundary = np.zeros(V, bool)
    on_boundary[a[bd]] = True; on_boundary[b[bd]] = True
    bsum = np.zeros_like(verts, dtype=np.float64)
    np.add.at(bsum, a[bd], verts[b[bd]]); np.add.at(bsum, b[bd], verts[a[bd]])
    k = np.maximum(val, 1)
    beta = (40.0 - (3.0 + 2.0 * np.cos(2 * np.pi / k)) ** 2) / (64.0 * k)
    even = (1 - k * beta)[:, None] * verts + beta[:, None] * nb_sum
    even[on_boundary] = 0.75 * verts[on_boundary] + 0.125 * bsum[on_boundary]
    even[val == 0] = verts[val == 0]
    m01, m12, m20 = V + inv[:F], V + inv[F:2 * F], V + inv[2 * F:]
    f = faces.astype(np.int64)
    kids = np.stack([np.stack([f[:, 0], m01, m20], 1), np.stack([m01, f[:, 1], m12], 1),
                     np.stack([m20, m12, f[:, 2]], 1), np.stack([m01, m12, m20], 1)], 1).reshape(-1, 3)
    return np.concatenate([even, odd], 0), kids


def subdivide_loop(verts, faces, iterations=3):
    """-> (vertices [V',3] float64, faces [4^it F, 3] int64), as trimesh.remesh.subdivide_loop."""
    v, f = np.asarray(verts, dtype=np.float64), np.asarray(faces)
    for _ in range(iterations):
        v, f = subdivide_loop_once(v, f)
    return v, f
