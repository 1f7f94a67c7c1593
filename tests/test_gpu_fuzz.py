"""Differential test on HOSTILE geometry: seeded random triangle soups built to hit every rule of the rasterizer at once — vertices
behind the camera or on it (w <= 0, w tiny), far outside the frustum, exactly on pixel centres and pixel corners (the tie rules R3 / R4),
non-finite coordinates, zero-area and repeated-index triangles, duplicated and reversed faces (the depth tie goes to the lower index),
resolutions from 1 x 1 up with odd widths.  The kernels' DISCRETE outputs (triangle ids, the antialias work queue) and the fp32 values
of `rast` must be bit-identical to the CPU checker's; nothing here is tuned to a mesh the step was built for."""
import numpy as np
import pytest
import torch

from util import mesh_scene, sort_items

pytestmark = pytest.mark.gpu


def hostile_soup(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 72)), int(rng.integers(1, 72))
    if seed % 7 == 0:
        H, W = int(rng.integers(1, 4)), int(rng.integers(1, 4))          # 1 .. 3 pixels a side
    nv, nf = int(rng.integers(3, 64)), int(rng.integers(1, 128))
    w = rng.uniform(0.4, 2.0, (n, nv, 1))
    xy = rng.uniform(-1.3, 1.3, (n, nv, 2))
    z = rng.uniform(-1.2, 1.2, (n, nv, 1))
    kind = rng.integers(0, 16, (n, nv))
    # exactly on a pixel centre / a pixel corner (in NDC; times w = 1 so that the clip value IS the NDC value)
    px, py = rng.integers(0, W + 1, (n, nv)), rng.integers(0, H + 1, (n, nv))
    centre = np.stack([(2.0 * px + 1.0) / W - 1.0, (2.0 * py + 1.0) / H - 1.0], -1)
    corner = np.stack([2.0 * px / W - 1.0, 2.0 * py / H - 1.0], -1)
    on_c, on_k = (kind == 0)[..., None], (kind == 1)[..., None]
    xy = np.where(on_c, centre, np.where(on_k, corner, xy))
    w = np.where(on_c | on_k, 1.0, w)
    w = np.where((kind == 2)[..., None], -w, w)                          # behind the camera
    w = np.where((kind == 3)[..., None], 0.0, w)                         # on it
    w = np.where((kind == 4)[..., None], 1e-7, w)                        # all but on it: NDC ~ 1e7
    xy = np.where((kind == 5)[..., None], xy * 60.0, xy)                 # far outside the frustum (large boxes, clamped to the screen)
    pos = np.concatenate([xy * w, z * w, w], 2).astype(np.float32)
    bad = kind == 6
    if seed % 3 == 0:                                                    # non-finite coordinates in a third of the scenes
        pos[bad, rng.integers(0, 4)] = [np.nan, np.inf, -np.inf][seed % 9 // 3]
    faces = rng.integers(0, nv, (nf, 3))
    k = max(1, nf // 6)
    faces[:k, 2] = faces[:k, 1]                                          # repeated index: zero area
    if nf > 3 * k:
        dup = faces[2 * k:3 * k].copy()                                  # duplicated faces: the depth tie goes to the lower index
        faces[k:2 * k] = dup[:, [0, 2, 1]] if seed % 2 else dup         # ... reversed ones in every other scene
    return torch.from_numpy(pos), torch.from_numpy(faces.astype(np.int32)), H, W


@pytest.mark.parametrize("seed", range(48))
def test_hostile_soup_is_rasterized_bit_identically(oracle, seed):
    from fmhr_amd import dr
    dev = torch.device("cuda:0")
    pos, tri, H, W = hostile_soup(seed)
    r_ref, _ = oracle.rasterize(None, pos, tri, (H, W))
    r_gpu, _ = dr.rasterize(dr.RasterizeGLContext(), pos.to(dev), tri.to(dev), resolution=(H, W))
    r_g = r_gpu.cpu()
    assert torch.equal(r_g[..., 3], r_ref[..., 3]), f"seed {seed} ({tuple(pos.shape)}, {H}x{W}): ids differ at {int((r_g[..., 3] != r_ref[..., 3]).sum())} px"
    assert torch.equal(r_g, r_ref), f"seed {seed}: values differ, max {float((r_g - r_ref).abs().max())}"
    assert bool(torch.isfinite(r_g).all())
    # the antialias analysis of the same scene: the same work queue (pixel pairs, edges, sides) and bit-identical alphas
    rng = np.random.default_rng(seed)
    color = torch.tensor(rng.uniform(0, 1, (pos.shape[0], H, W, 3)), dtype=torch.float32)
    a_ref, it_ref, al_ref = oracle.antialias(color, r_ref, pos, tri, return_items=True)
    a_gpu, it_gpu, al_gpu, overflow = dr.antialias_debug_items(color.to(dev), r_gpu, pos.to(dev), tri.to(dev))
    assert overflow == 0
    si, sa = sort_items(it_gpu, al_gpu)
    assert si.shape[0] == it_ref.shape[0], f"seed {seed}: {si.shape[0]} antialias items, the checker has {it_ref.shape[0]}"
    if it_ref.shape[0]:
        assert torch.equal(si, it_ref.long()), f"seed {seed}: antialias work items differ"
        assert torch.equal(sa, al_ref), f"seed {seed}: alphas differ"
    assert float((a_gpu.cpu() - a_ref).abs().max()) < 1e-6


def hostile_cameras(oracle, dims, seed):
    """a synthetic HAM scene (the mesh the step was built for) seen by cameras it was not: per view one of — half off-screen, mostly
    off-screen, zoomed in 6 x (large boxes, most of them clipped by the screen), zoomed out 12 x (the whole mesh in a few pixels: nearly
    every triangle sub-pixel), mirrored (every winding flipped), and pushed INTO the mesh (w <= 0 for about a third of the vertices:
    their triangles are dropped, rule R1).  Targets are rendered by the checker with the same cameras."""
    from fmhr_amd import synthetic as S
    from checker_scenes import oracle_forward_fn
    sc = mesh_scene(dims, seed=seed)
    rng = np.random.default_rng(500 + seed)
    proj = sc["proj"].clone()
    v4 = torch.cat([sc["vertices"], torch.ones(sc["V"], 1)], 1)
    for b in range(sc["views"]):
        kind = int(rng.integers(0, 6))
        if kind == 0:
            proj[b, :, 0] += 0.9 * proj[b, :, 3]
        elif kind == 1:
            proj[b, :, 1] -= 1.15 * proj[b, :, 3]
        elif kind == 2:
            proj[b, :, :2] *= 6.0
        elif kind == 3:
            proj[b, :, :2] *= 0.08
        elif kind == 4:
            proj[b, :, 0] *= -1.0
        else:
            w = (v4 @ sc["w2c"][b] @ proj[b])[:, 3]
            proj[b, 3, 3] -= float(torch.quantile(w, 0.33))
    sc["proj"] = proj.contiguous()
    with torch.no_grad():
        tg = S.make_targets(sc, oracle_forward_fn(oracle, sc))
    sc["delta"] = torch.tensor(rng.standard_normal((sc["V"], 3)) * 2e-4, dtype=torch.float32)
    sc["albedo"] = (sc["albedo0"] + torch.tensor(rng.uniform(-0.1, 0.1, (sc["V"], 3)), dtype=torch.float32))
    tg["mask"][:, : sc["H"] // 5] = 0
    sc.update(tg)
    sc["elm"] = float(oracle.edge_length_mean(sc["vertices"], sc["faces"]))
    return sc


# (views, H, W, rings, segments): a micro-triangle mesh at 64 x 64 and 128 x 128 (the binned rasterizer) and a coarse one (atomic keys)
CAMERA_CASES = [(d, s) for d in ((3, 64, 64, 24, 32), (3, 96, 80, 12, 16), (4, 128, 128, 32, 48)) for s in range(6)]


@pytest.mark.parametrize("case", CAMERA_CASES, ids=lambda c: "%dx%dx%d_s%d" % (c[0][0], c[0][1], c[0][2], c[1]))
def test_hostile_cameras_through_the_fused_step(oracle, case):
    """the WHOLE step (mesh_sfs_optim.py:253-310) under those cameras, held to the bars of every other fused-step case: `rast`
    bit-identical to the fp32 checker, the six losses and both gradients within 1e-4 of fp64 values on the fp32 decisions"""
    from test_gpu_ham import _step_vs_oracle
    dims, seed = case
    sc = hostile_cameras(oracle, dims, seed)
    ld, v = _step_vs_oracle(oracle, sc, "hostile cameras %s seed %d" % (dims, seed))
    assert all(np.isfinite(x) for x in ld.values())


@pytest.mark.parametrize("dims", [(3, 64, 64, 24, 32), (3, 96, 80, 12, 16), (2, 64, 192, 40, 56)], ids=lambda d: "%dx%dx%d" % d[:3])
def test_one_workspace_follows_a_different_hostile_scene_every_step(oracle, dims):
    """everything the step keeps BETWEEN steps (segment state words, bin cursors, owner marks, coverage entries, the sliver workers' queue,
    the antialias lists) under the worst a caller can do to it: a new shape of the mesh, new cameras (views going empty and coming back,
    triangles switching between the small-box and the large-box path), new targets — every step on the SAME workspace; `rast` bit-identical
    to the fp32 checker at the step's own positions, losses and gradients against its fp32 evaluation (a stale entry shows as O(1)).
    (tools/experiments/r05_fuzz_stateful.py is the same loop over 40 scenes x 8 steps.)"""
    from fmhr_amd.ham import HamStage2
    from util import rel_err
    dev = torch.device("cuda:0")
    hs, faces = None, None
    for t in range(6):
        sc = hostile_cameras(oracle, dims, 7000 + t)
        if hs is None:
            faces = sc["faces"]
            hs = HamStage2(faces.to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
        assert torch.equal(sc["faces"], faces)                          # one topology; the vertices, cameras and targets change
        g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
        _, gd, ga = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
        ld, v = hs.loss_dict(), hs.views()
        d = sc["delta"].clone().requires_grad_(True)
        a = sc["albedo"][None].clone().requires_grad_(True)
        ref = oracle.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"], sc["w2c"], sc["proj"], sc["img"], sc["mask"], sc["valid_mask"], faces,
                                       (sc["H"], sc["W"]), sc["weights"], hs.edge_len_mean, pos_value=v["pos"].cpu())
        ref["loss"].backward()
        assert torch.equal(v["rast"].cpu(), ref["rast"]), (dims, t)
        for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"):
            assert abs(ld[k] - float(ref[k])) <= 1e-3 * max(abs(float(ref[k])), 1e-6), (dims, t, k, ld[k], float(ref[k]))
        assert rel_err(gd, d.grad) < 1e-3 and rel_err(ga, a.grad[0]) < 1e-3, (dims, t)
