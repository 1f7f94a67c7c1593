"""Deterministic synthetic HAM workloads: MANO-like open-wrist blob mesh, Fibonacci-sphere cameras with the
reference's projection convention (/root/reference/get_data.py:64-73,96-97), and hyper-parameters of conf/demo_sfs.conf.

Host-side numpy only (no rendering here): targets are rendered by whoever owns a renderer (the HIP path in bench/tests,
the CPU oracle in CPU tests) through `make_targets`.
"""
import math

import numpy as np
import torch

# /root/reference/conf/demo_sfs.conf:16-29
DEMO_WEIGHTS = dict(sfs=50.0, albedo=1.0, lap=2000.0, mask=1000.0, edge=500000.0, delta=50000.0)
DEMO_LR = dict(lr=1e-4, albedo_lr=5e-3, sh_lr=5e-3)
# half extents of the MANO template bounding box
SEMI_AXES = (0.0965, 0.0306, 0.0838)

CONFIGS = {
    # name: (views, H, W, rings, segments, subdivisions[, hands[, frames]]).  rings = "mano": the tapered disc base of `mano_disc_base`
    # with EXACTLY MANO's counts (V=778 / F=1538 / E=2315, 16 boundary edges: mano/TEMPLATE_RIGHT.obj); three 1:4 subdivisions give
    # V=49281 / F=98432 / E=147712 — the sizes /root/reference/train_mlp.py:150-154 hard-codes — with MANO-like bounded valence and
    # the face/vertex order contract of SURVEY.md Appendix E.  (A plain rings x segments UV blob, e.g. 21 x 37, has V=778 / F=1517
    # and subdivides to 48693 / 97088: it stays available for the parity tests' small shapes.)
    "C1": (1, 256, 256, "mano", 0, 0),
    "C2": (8, 512, 334, "mano", 0, 0),
    "C3": (16, 512, 512, "mano", 0, 3),
    "metric": (48, 512, 512, "mano", 0, 3),
    "C4": (48, 1024, 1024, "mano", 0, 3),
    # experiments only (the view-stride question): the metric workload with eight more rows, so that the views' pixel
    # arrays lie 4.06 MiB apart instead of exactly 4 MiB
    "metric_h520": (48, 520, 512, "mano", 0, 3),
    # BASELINE.json configs[4]: two hands (each subdivided on its own, concatenated as mesh_sfs_optim.py:76-87,100-111 does),
    # 32 views x 8 frames of 334 x 512 (conf/ih_sfsseq.conf:11-13); a frame is an independent (delta, albedo) problem on the shared
    # topology (fmhr_amd.dist.shard_frames_views)
    "C5": (32, 512, 334, "mano", 0, 3, 2, 8),
    # the reference's own demo run (conf/demo_sfs.conf:5-7,24: a capture of 16 views, 1280 x 1024, two hands, minibatches of 8): bench.py
    # drives it through the product's loop (LOOP_BATCH below: fmhr_amd.pipeline.Stage2Loop — a fresh permutation per epoch, the batch's views
    # selected inside the step and, on N ranks, sharded: `--config demo --gpus 8` is one view per GPU and step)
    "demo": (16, 1024, 1280, "mano", 0, 3, 2),
    # ONE fixed batch of it (8 views, no permutation): the bare step at that size
    "demo_batch": (8, 1024, 1280, "mano", 0, 3, 2),
}
# configs whose step is a MINIBATCH of the capture drawn by the loop (mesh_sfs_optim.py:247-260): name -> batch
LOOP_BATCH = {"demo": 8}


TH_MAX = 0.86 * math.pi


def _blob_surface(th, ph, seed, bump):
    """points of the bumpy ellipsoid at spherical parameters (th from the pole on the long axis, ph around it)."""
    rng = np.random.default_rng(seed)
    disp = np.zeros_like(th * ph)
    for _ in range(6):                     # seeded low-frequency radial displacement (concavities / self-occlusion)
        kt, kp = rng.integers(1, 4), rng.integers(1, 5)
        a, p0, p1 = rng.uniform(0.3, 1.0), rng.uniform(0, 2 * math.pi), rng.uniform(0, 2 * math.pi)
        disp = disp + a * np.sin(kt * th + p0) * np.cos(kp * ph + p1) * np.sin(th)
    disp = 1.0 + bump * disp / 3.0
    d = np.stack([np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph), np.sin(th) * np.cos(ph)], -1) * disp[..., None]
    return d * np.array(SEMI_AXES)


def subdivide(faces, V):
    """1:4 midpoint subdivision with the order contract of SURVEY.md Appendix E (repose.py:14-24 relies on it): original
    vertices keep their indices, edge midpoints are appended in unique-edge order, each parent [f0,f1,f2] becomes the four
    consecutive children [f0,m01,m02], [m01,f1,m12], [m02,m12,f2], [m01,m12,m02].  Returns (faces', edges [E,2], V')."""
    f = faces.astype(np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    e.sort(axis=1)
    uniq, inv = np.unique(e[:, 0] * V + e[:, 1], return_inverse=True)
    edges = np.stack([uniq // V, uniq % V], 1)
    Fn = f.shape[0]
    m01, m12, m20 = V + inv[:Fn], V + inv[Fn:2 * Fn], V + inv[2 * Fn:]
    kids = np.stack([np.stack([f[:, 0], m01, m20], 1), np.stack([m01, f[:, 1], m12], 1),
                     np.stack([m20, m12, f[:, 2]], 1), np.stack([m01, m12, m20], 1)], 1).reshape(-1, 3)
    return kids.astype(np.int32), edges, V + edges.shape[0]


def blob_mesh(rings, segs, seed=0, bump=0.10, subdiv=0):
    """Open-wrist UV blob: closed top pole, open bottom => disc topology like MANO.  Base: V = rings*segs+1,
    F = segs + 2*segs*(rings-1); then `subdiv` 1:4 subdivisions whose new vertices are put on the analytic surface (midpoints in
    parameter space) — the counts and index layout of a Loop-subdivided MANO without needing trimesh."""
    i = np.arange(1, rings + 1, dtype=np.float64)[:, None]
    j = np.arange(segs, dtype=np.float64)[None, :]
    th = (TH_MAX * i / rings) * np.ones((1, segs))
    ph = (2 * math.pi * j / segs) * np.ones((rings, 1))
    th = np.concatenate([[0.0], th.reshape(-1)])
    ph = np.concatenate([[0.0], ph.reshape(-1)])
    faces = []
    ring0 = 1 + np.arange(segs)
    nxt = 1 + (np.arange(segs) + 1) % segs
    faces.append(np.stack([np.zeros(segs, dtype=np.int64), ring0, nxt], 1))
    for r in range(1, rings):
        a0, b0 = (r - 1) * segs, r * segs
        faces.append(np.stack([a0 + ring0, b0 + ring0, b0 + nxt], 1))
        faces.append(np.stack([a0 + ring0, b0 + nxt, a0 + nxt], 1))
    faces = np.concatenate(faces, 0).astype(np.int32)
    th, ph, faces = _subdivide_params(th, ph, faces, subdiv)
    verts = _blob_surface(th[:, None], ph[:, None], seed, bump)[:, 0, :]
    return verts.astype(np.float32), faces.astype(np.int32)


def mano_disc_base():
    """Tapered open-wrist blob with exactly the counts of the MANO template (V=778, F=1538, E=2315; disc topology with a
    16-edge wrist loop): a pole + 29 latitude rings whose segment counts follow the ring circumference (max 38) down to 16 at the
    open wrist, neighbouring rings zipped by angle.  F = 2(V-1) - B holds for any such ring profile; the profile below sums to
    777 ring vertices.  Returns (th [V], ph [V], faces [F,3] int32): parameters on the blob surface, so that subdivision can put
    new vertices on the analytic surface."""
    rings, wrist = 29, 16
    sin = np.sin(TH_MAX * np.arange(1, rings + 1) / rings)
    k = wrist / sin[-1]
    counts = np.maximum(6, np.rint(k * sin)).astype(np.int64)
    counts[-1] = wrist
    # hit 777 exactly by trimming / padding the widest rings one vertex at a time (deterministic order)
    order = np.argsort(-sin[:-1], kind="stable")
    i = 0
    while counts.sum() != 777:
        counts[order[i % order.size]] += 1 if counts.sum() < 777 else -1
        i += 1
    th = [0.0]
    ph = [0.0]
    start = [1]
    for r, c in enumerate(counts):
        th += [TH_MAX * (r + 1) / rings] * int(c)
        ph += list(2 * math.pi * np.arange(c) / c)
        start.append(start[-1] + int(c))
    faces = [[0, 1 + j, 1 + (j + 1) % int(counts[0])] for j in range(int(counts[0]))]
    for r in range(rings - 1):
        m, kk = int(counts[r]), int(counts[r + 1])
        u0, l0 = start[r], start[r + 1]
        i = j = 0
        for _ in range(m + kk):
            # next angle on either ring, in exact integer arithmetic: (i+1)/m vs (j+1)/kk
            lower = j < kk and (i == m or (j + 1) * m <= (i + 1) * kk)
            if lower:
                faces.append([u0 + i % m, l0 + j % kk, l0 + (j + 1) % kk])
                j += 1
            else:
                faces.append([u0 + i % m, l0 + j % kk, u0 + (i + 1) % m])
                i += 1
    return np.array(th), np.array(ph), np.array(faces, dtype=np.int32)


def mano_disc_mesh(seed=0, bump=0.10, subdiv=0):
    """`mano_disc_base` on the bumpy ellipsoid, subdivided `subdiv` times 1:4 (new vertices on the analytic surface)."""
    th, ph, faces = mano_disc_base()
    th, ph, faces = _subdivide_params(th, ph, faces, subdiv)
    verts = _blob_surface(th[:, None], ph[:, None], seed, bump)[:, 0, :]
    return verts.astype(np.float32), faces.astype(np.int32)


def _subdivide_params(th, ph, faces, subdiv):
    for _ in range(subdiv):
        V = th.shape[0]
        faces, edges, _ = subdivide(faces, V)
        a, b = edges[:, 0], edges[:, 1]
        pa, pb = ph[a], ph[b]
        # midpoint in parameter space; the pole (index 0, th = 0) has no meaningful ph: take the other end's
        pa = np.where(th[a] == 0, pb, pa)
        pb = np.where(th[b] == 0, pa, pb)
        dphi = (pb - pa + math.pi) % (2 * math.pi) - math.pi
        th = np.concatenate([th, 0.5 * (th[a] + th[b])])
        ph = np.concatenate([ph, pa + 0.5 * dphi])
    return th, ph, faces


def two_hands(verts, faces, gap=0.075):
    """Second hand = the first mirrored in y and shifted (orientation kept by swapping two corners), concatenated as the reference
    concatenates its per-hand subdivisions (mesh_sfs_optim.py:76-87): vertices [2V,3], faces [2F,3] with the second hand's indices
    offset by V — two connected components, two boundary loops."""
    v2 = verts * np.array([1.0, -1.0, 1.0], dtype=verts.dtype) + np.array([0.0, gap, 0.0], dtype=verts.dtype)
    f2 = faces[:, [0, 2, 1]] + verts.shape[0]
    return np.concatenate([verts, v2], 0).astype(np.float32), np.concatenate([faces, f2], 0).astype(np.int32)


def fibonacci_dirs(n):
    k = np.arange(n) + 0.5
    z = 1 - 2 * k / n
    r = np.sqrt(np.maximum(0, 1 - z * z))
    ph = k * math.pi * (3 - math.sqrt(5))
    return np.stack([r * np.cos(ph), r * np.sin(ph), z], 1)


def cameras(verts, n, H, W, coverage=0.25, dist=0.6):
    """n OpenCV-convention cameras on a Fibonacci sphere looking at the centroid.  Returns (w2c, proj) float32 [n,4,4]
    stored TRANSPOSED for the reference's row-vector convention (get_data.py:96-97), focal chosen per view so the
    projected mesh covers ~`coverage` of the image."""
    c = verts.mean(0).astype(np.float64)
    w2cs, projs = [], []
    for d in fibonacci_dirs(n):
        eye = c + dist * d
        fwd = (c - eye) / np.linalg.norm(c - eye)
        up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.95 else np.array([0.0, 1.0, 0.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], 0)          # world -> camera rows (x right, y down, z forward)
        t = -R @ eye
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R, t
        pc = (verts.astype(np.float64) @ R.T + t)
        xy = pc[:, :2] / pc[:, 2:3]
        ev = np.linalg.eigvalsh(np.cov(xy.T))
        area_unit = math.pi * 4.0 * math.sqrt(max(ev[0], 1e-12) * max(ev[1], 1e-12))   # ellipse with radii 2 sigma
        f = math.sqrt(coverage * H * W / area_unit)
        ctr = xy.mean(0)
        cx, cy = W / 2.0 - f * ctr[0], H / 2.0 - f * ctr[1]
        proj = np.zeros((4, 4))
        proj[0, 0], proj[0, 2] = f / (W / 2.0), cx / (W / 2.0) - 1.0
        proj[1, 1], proj[1, 2] = f / (H / 2.0), cy / (H / 2.0) - 1.0
        proj[2, 3] = -0.1
        proj[3, 2] = 1.0
        w2cs.append(w2c.T)
        projs.append(proj.T)
    return np.stack(w2cs).astype(np.float32), np.stack(projs).astype(np.float32)


def make_scene(views, H, W, rings, segs, subdiv=0, seed=0, coverage=0.25, hands=1):
    """Everything the stage-2 loop needs except the rendered targets (see make_targets).  rings = "mano": the exact-count disc
    base (segs ignored); hands = 2: the two-component mesh of the reference's two-hand captures."""
    rng = np.random.default_rng(seed)
    if rings == "mano":
        verts, faces = mano_disc_mesh(seed, subdiv=subdiv)
    else:
        verts, faces = blob_mesh(rings, segs, seed, subdiv=subdiv)
    if hands == 2:
        verts, faces = two_hands(verts, faces)
    w2c, proj = cameras(verts, views, H, W, coverage)
    V = verts.shape[0]
    sh_true = np.concatenate([np.full((views, 1), 0.8), 0.1 * rng.standard_normal((views, 8))], 1).astype(np.float32)
    albedo_true = rng.uniform(0.3, 0.9, (V, 3)).astype(np.float32)
    verts_true = (verts + 0.0005 * rng.standard_normal(verts.shape)).astype(np.float32)
    sh0 = (sh_true + 0.02 * rng.standard_normal(sh_true.shape)).astype(np.float32)
    albedo0 = np.full((V, 3), 0.6, dtype=np.float32)
    t = torch.from_numpy
    return dict(views=views, H=H, W=W, V=V, F=faces.shape[0], hands=hands,
                vertices=t(verts), faces=t(faces), w2c=t(w2c), proj=t(proj),
                vertices_true=t(verts_true), albedo_true=t(albedo_true), sh_true=t(sh_true),
                albedo0=t(albedo0), sh0=t(sh0), weights=dict(DEMO_WEIGHTS), lr=dict(DEMO_LR))


def scene_from_config(name, seed=0, views=None):
    v, H, W, r, c, sd = CONFIGS[name][:6]
    hands = CONFIGS[name][6] if len(CONFIGS[name]) > 6 else 1
    sc = make_scene(views or v, H, W, r, c, sd, seed, hands=hands)
    sc["frames"] = CONFIGS[name][7] if len(CONFIGS[name]) > 7 else 1
    return sc


def make_targets(scene, forward_fn, device=None):
    """Render the targets with the caller's renderer.  forward_fn(vertices[V,3], albedo[V,3], sh[n,9], mask[n,H,W]) must
    return dict(tmp_img [n,H,W,3], pred_mask [n,H,W], rast [n,H,W,4]) — the stage-2 forward of the reference.
    img = shaded render of the perturbed copy; mask = its coverage; valid_mask = AA silhouette of the START mesh
    (mesh_sfs_optim.py:146,158)."""
    n, H, W = scene["views"], scene["H"], scene["W"]
    dev = device or scene["vertices"].device
    ones = torch.ones(n, H, W, dtype=torch.float32, device=dev)
    tgt = forward_fn(scene["vertices_true"].to(dev), scene["albedo_true"].to(dev), scene["sh_true"].to(dev), ones)
    img = tgt["tmp_img"].detach().clone()
    mask = (tgt["rast"][..., 3] > 0).to(torch.float32)
    start = forward_fn(scene["vertices"].to(dev), scene["albedo0"].to(dev), scene["sh0"].to(dev), ones)
    valid_mask = start["pred_mask"].detach().clone()
    cov = float((start["rast"][..., 3] > 0).to(torch.float32).mean())
    return dict(img=img, mask=mask, valid_mask=valid_mask, coverage=cov)
