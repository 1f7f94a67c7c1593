"""GPU parity of the nvdiffrast-shaped ops (fmhr_amd.dr, through the C ABI) against the CPU oracle.
Discrete outputs (triangle ids, antialias work items) must be bit-exact against the fp32 oracle; values within 1e-4
(per-tensor max|a-b|/max|b|), the tolerance BASELINE.json:north_star states."""
import numpy as np
import pytest
import torch

from util import mesh_scene, random_scene, rel_err, sort_items

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dr():
    from fmhr_amd import dr as _dr
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU path"
    return _dr


def scenes():
    out = []
    for seed in range(4):
        pos, tri, rng = random_scene(seed, n=2, H=40, W=36, nv=30, nf=50)
        out.append((f"soup{seed}", pos, tri, 40, 36))
    sc = mesh_scene((3, 96, 80, 12, 16), seed=1)
    out.append(("blob_small", sc["pos"], sc["faces"], 96, 80))
    sc = mesh_scene((2, 512, 334, 24, 32), seed=0)          # config C2 geometry (W = 334: no alignment assumptions)
    out.append(("C2", sc["pos"], sc["faces"], 512, 334))
    sc = mesh_scene((2, 256, 256, 96, 128), seed=2)         # micro-triangle regime
    out.append(("micro", sc["pos"], sc["faces"], 256, 256))
    # two hands: two connected components, each with its own boundary loop, concatenated as mesh_sfs_optim.py:76-87 does; they
    # occlude each other in some views (inter-component depth order and silhouette pairs), W = 334 (conf/ih_sfsseq.conf:11)
    sc = mesh_scene((3, 256, 334, 12, 16, 1, 2), seed=3)
    out.append(("two_hands", sc["pos"], sc["faces"], 256, 334))
    return out


@pytest.mark.parametrize("case", scenes(), ids=lambda c: c[0])
def test_rasterize_interpolate_antialias_forward(dr, oracle, case):
    name, pos, tri, H, W = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    r_ref, _ = oracle.rasterize(None, pos, tri, (H, W))
    r_gpu, db = dr.rasterize(dr.RasterizeGLContext(), pos.to(dev), tri.to(dev), resolution=(H, W))
    assert db.shape == r_gpu.shape
    r_g = r_gpu.cpu()
    assert torch.equal(r_g[..., 3], r_ref[..., 3]), f"{name}: triangle ids differ at {(r_g[..., 3] != r_ref[..., 3]).sum()} px"
    assert torch.equal(r_g, r_ref), f"{name}: rast must be bit-identical to the fp32 oracle (max diff {(r_g - r_ref).abs().max()})"
    assert (r_ref[..., 3] > 0).float().mean() > 0.02
    # double-precision oracle: same ids almost everywhere, values within tolerance
    r64, _ = oracle.rasterize(None, pos.double(), tri, (H, W))
    same = r64[..., 3] == r_g[..., 3].double()
    assert same.float().mean() > 0.995
    assert (r64[..., :3] - r_g[..., :3].double())[same].abs().max() < 2e-3   # fp32 cancellation of the homogeneous edge functions
    # interpolate, C = 7 (the HAM call) and C = 1
    V = pos.shape[1]
    for C, an in ((7, pos.shape[0]), (1, 1), (4, 1)):
        attr = torch.tensor(rng.standard_normal((an, V, C)), dtype=torch.float32)
        o_ref, _ = oracle.interpolate(attr, r_ref, tri)
        o_gpu, _ = dr.interpolate(attr.to(dev), r_gpu, tri.to(dev))
        assert rel_err(o_gpu, o_ref) < 1e-6
    ones, _ = dr.interpolate(torch.ones(1, V, 1, device=dev), r_gpu, tri.to(dev))
    assert set(np.unique(ones.cpu().numpy().round(5))) <= {0.0, 1.0}
    # antialias: identical work queue, values within tolerance
    for C in (1, 3, 2):
        color = torch.tensor(rng.uniform(0, 1, (pos.shape[0], H, W, C)), dtype=torch.float32)
        a_ref, it_ref, al_ref = oracle.antialias(color, r_ref, pos, tri, return_items=True)
        a_gpu, it_gpu, al_gpu, overflow = dr.antialias_debug_items(color.to(dev), r_gpu, pos.to(dev), tri.to(dev))
        assert overflow == 0
        si, sa = sort_items(it_gpu, al_gpu)
        assert si.shape[0] == it_ref.shape[0] and it_ref.shape[0] > 0
        assert torch.equal(si, it_ref.long()), f"{name}: antialias work items differ"
        assert torch.equal(sa, al_ref), f"{name}: alphas must be bit-identical"
        assert rel_err(a_gpu, a_ref) < 1e-6
        a2 = dr.antialias(color.to(dev), r_gpu, pos.to(dev), tri.to(dev))
        assert rel_err(a2, a_ref) < 1e-6


@pytest.mark.parametrize("case", scenes()[:5] + scenes()[7:], ids=lambda c: c[0])
def test_backward_parity(dr, oracle, case):
    """the composed rasterize -> interpolate -> antialias chain, gradients w.r.t. pos / attr / color vs the oracle
    (fp32 twin and fp64)."""
    name, pos, tri, H, W = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    n, V = pos.shape[0], pos.shape[1]
    C = 5
    attr0 = torch.tensor(rng.standard_normal((n, V, C)), dtype=torch.float32)
    wt = torch.tensor(rng.standard_normal((n, H, W, C)), dtype=torch.float32)

    def run(mod, p, a, w_, tri_, ctx):
        p = p.clone().requires_grad_(True)
        a = a.clone().requires_grad_(True)
        rast, _ = mod.rasterize(ctx, p, tri_, resolution=(H, W))
        feat, _ = mod.interpolate(a, rast, tri_)
        aa = mod.antialias(feat, rast, p, tri_)
        loss = (aa * w_).sum() + (feat * w_).sum() * 0.5
        loss.backward()
        return loss.detach(), p.grad, a.grad

    l_ref, gp_ref, ga_ref = run(oracle, pos, attr0, wt, tri, None)
    l64, gp64, ga64 = run(oracle, pos.double(), attr0.double(), wt.double(), tri, None)
    l_gpu, gp_gpu, ga_gpu = run(dr, pos.to(dev), attr0.to(dev), wt.to(dev), tri.to(dev), dr.RasterizeGLContext())
    assert rel_err(l_gpu, l_ref) < TOL
    assert rel_err(gp_gpu, gp_ref) < TOL, name
    assert rel_err(ga_gpu, ga_ref) < TOL, name
    # against fp64 the fp32 decisions may differ at a handful of marginal pixels; compare where the oracle twins agree
    if rel_err(gp_ref, gp64) < 1e-3:
        assert rel_err(gp_gpu, gp64) < 2e-3
        assert rel_err(ga_gpu, ga64) < 2e-3


def test_feature_rasterisation_c30(dr, oracle):
    """the second consumer of the primitives: train_mlp.py:165-205 rasterizes once and interpolates a
    30-channel per-vertex feature (shared over the views, attr [1,V,30]) — forward and the gradients w.r.t. feature and
    barycentrics (-> pos) against the oracle."""
    dev = torch.device("cuda:0")
    sc = mesh_scene((2, 128, 112, 24, 32), seed=4)
    pos, tri, H, W = sc["pos"], sc["faces"], 128, 112
    V, C = pos.shape[1], 30
    rng = np.random.default_rng(5)
    feat0 = torch.tensor(rng.standard_normal((1, V, C)), dtype=torch.float32)
    wt = torch.tensor(rng.standard_normal((2, H, W, C)), dtype=torch.float32)

    def run(mod, p, a, w_, tri_, ctx):
        p = p.clone().requires_grad_(True)
        a = a.clone().requires_grad_(True)
        rast, _ = mod.rasterize(ctx, p, tri_, resolution=(H, W))
        feat, _ = mod.interpolate(a, rast, tri_)
        (feat * w_).sum().backward()
        return rast.detach(), feat.detach(), p.grad, a.grad

    r_ref, f_ref, gp_ref, ga_ref = run(oracle, pos, feat0, wt, tri, None)
    r_gpu, f_gpu, gp_gpu, ga_gpu = run(dr, pos.to(dev), feat0.to(dev), wt.to(dev), tri.to(dev), dr.RasterizeGLContext())
    assert torch.equal(r_gpu.cpu(), r_ref)
    assert f_gpu.shape == (2, H, W, C) and rel_err(f_gpu, f_ref) < 1e-6
    assert rel_err(ga_gpu, ga_ref) < TOL and rel_err(gp_gpu, gp_ref) < TOL
    _, _, gp64, ga64 = run(oracle, pos.double(), feat0.double(), wt.double(), tri, None)
    if rel_err(gp_ref, gp64) < 1e-3:
        assert rel_err(gp_gpu, gp64) < 2e-3 and rel_err(ga_gpu, ga64) < 2e-3


def test_edge_cases(dr):
    dev = torch.device("cuda:0")
    ctx = dr.RasterizeGLContext()
    pos = torch.tensor([[[0, 0, 0, 1], [0.5, 0, 0, 1], [0, 0.5, 0, 1.]]], device=dev)
    tri0 = torch.zeros(0, 3, dtype=torch.int32, device=dev)
    rast, _ = dr.rasterize(ctx, pos, tri0, (5, 7))
    assert rast.shape == (1, 5, 7, 4) and (rast == 0).all()
    out, _ = dr.interpolate(torch.ones(1, 3, 2, device=dev), rast, tri0)
    assert (out == 0).all()
    aa = dr.antialias(torch.rand(1, 5, 7, 3, device=dev), rast, pos, tri0)
    assert aa.shape == (1, 5, 7, 3)
    # one huge triangle covering the whole 300x200 screen goes through the large-triangle queue
    big = torch.tensor([[[-4, -4, 0, 1], [4, -4, 0, 1], [0, 4, 0, 1.]]], device=dev)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=dev)
    rast, _ = dr.rasterize(ctx, big, tri, (200, 300))
    assert (rast[..., 3] == 1).all()
    assert torch.allclose(rast[..., 0] + rast[..., 1], 1 - (1 - rast[..., 0] - rast[..., 1]), atol=1e-6)
    # behind-the-camera vertex: whole triangle dropped
    bad = big.clone()
    bad[0, 0, 3] = -1.0
    rast, _ = dr.rasterize(ctx, bad, tri, (16, 16))
    assert (rast == 0).all()
    with pytest.raises(RuntimeError):
        dr.rasterize(ctx, big, tri.long(), (16, 16))
    with pytest.raises(RuntimeError):
        dr.interpolate(torch.ones(3, 3, 2, device=dev), rast, tri)


def test_topology_and_determinism(dr, oracle):
    dev = torch.device("cuda:0")
    sc = mesh_scene((2, 128, 128, 24, 32), seed=3)
    tri = sc["faces"]
    ctx = dr.RasterizeGLContext()
    opp = ctx.opp_table(tri.to(dev), sc["V"]).cpu()
    assert torch.equal(opp, oracle.topology_build(tri, sc["V"]))
    # non-manifold fan: three faces on one edge -> the two lowest codes win
    fan = torch.tensor([[0, 1, 2], [1, 0, 3], [0, 1, 4]], dtype=torch.int32)
    assert torch.equal(dr.RasterizeGLContext().opp_table(fan.to(dev), 5).cpu(), oracle.topology_build(fan, 5))
    r1, _ = dr.rasterize(ctx, sc["pos"].to(dev), tri.to(dev), (128, 128))
    r2, _ = dr.rasterize(ctx, sc["pos"].to(dev), tri.to(dev), (128, 128))
    assert torch.equal(r1, r2), "rasterize must be run-to-run deterministic"
