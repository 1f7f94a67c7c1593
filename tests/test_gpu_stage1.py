"""GPU parity of the fused stage-1 iteration (fmhr_amd.stage1.HamStage1, include/fmhr_ham.h: fmhr_stage1_*) against the op-by-op loop
body of mesh_sfs_optim.py:205-236 — `pipeline.stage1_loss` on the product's generic ops (autograd), and the same code on the CPU
checker — for the loss, the albedo gradient and the SH gradient; view subsets; the fused Adam update; the metric mesh."""
import pytest
import torch

from util import rel_err
from test_next_rows import _tiny_capture

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _scene(dev, views=6, H=128, W=96, seed=5):
    from fmhr_amd import synthetic as S
    from fmhr_amd.ham import render_forward
    sc = S.make_scene(views, H, W, 21, 37, 1, seed=seed)
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous()
    tg = S.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    return sc, faces, w2c, proj, tg


def _reference(pipeline, ops, vt, faces, albedo, sh, w2c, proj, img, mask, res, idx=None):
    a = albedo.detach().clone().requires_grad_(True)
    s = sh.detach().clone().requires_grad_(True)
    sel = slice(None) if idx is None else idx
    loss = pipeline.stage1_loss(ops, ops.RasterizeGLContext(), vt, faces, a, s[sel], w2c[sel], proj[sel], img[sel], mask[sel], res, 50.0)
    loss.backward()
    return loss.detach(), a.grad[0], s.grad


def test_fused_stage1_step_matches_the_op_by_op_loop_body():
    from fmhr_amd import pipeline
    from fmhr_amd.stage1 import HamStage1
    dev = torch.device("cuda:0")
    sc, faces, w2c, proj, tg = _scene(dev)
    vt = sc["vertices"].to(dev).contiguous()
    res = (sc["H"], sc["W"])
    ops = pipeline.product_ops()
    torch.manual_seed(3)
    albedo = (sc["albedo0"][None].to(dev) + 0.05 * torch.randn(1, sc["V"], 3, device=dev)).contiguous()
    sh = (sc["sh0"].to(dev) + 0.05 * torch.randn(sc["views"], 9, device=dev)).contiguous()
    hs = HamStage1(vt, faces, w2c, proj, tg["img"], tg["mask"], res, 50.0, chunk=4)
    assert hs.rows % 256 == 0 and hs.entries >= 3 * int(hs.view_valid.sum())
    lv, ga, gs = hs.step(albedo, sh)
    loss = lv.sum()
    lr, gar, gsr = _reference(pipeline, ops, vt, faces, albedo, sh, w2c, proj, tg["img"], tg["mask"], res)
    assert abs(float(loss) - float(lr)) < TOL * abs(float(lr))
    assert rel_err(ga, gar) < TOL and rel_err(gs, gsr) < TOL
    # a batch of some of the views (:199-203): value, gradients; the other views' SH rows get a zero gradient
    idx = torch.tensor([4, 1, 2], device=dev)
    lv, ga, gs = hs.step(albedo, sh, idx=idx)
    loss = lv.sum()
    lr, gar, gsr = _reference(pipeline, ops, vt, faces, albedo, sh, w2c, proj, tg["img"], tg["mask"], res, idx=idx)
    assert abs(float(loss) - float(lr)) < TOL * abs(float(lr))
    assert rel_err(ga, gar) < TOL and rel_err(gs, gsr) < TOL
    assert float(gs[torch.tensor([0, 3, 5], device=dev)].abs().max()) == 0.0
    # the step is deterministic (gather-only, fixed summation order)
    ga_first = ga.clone()
    l2, ga2, gs2 = hs.step(albedo, sh, idx=idx)
    assert torch.equal(ga2, ga_first) and float(l2.sum()) == float(loss)
    # the selection as the C ABI takes it — an int32 mask over the views, the form run_ham's loop passes (fmhr_amd.pipeline.PermChunks:
    # the masks of 64 epochs' batches in one transfer, no index kernels in front of a step): the same bits
    sel = torch.zeros(sc["views"], dtype=torch.int32, device=dev)
    sel[idx] = 1
    l3, ga3, gs3 = hs.step(albedo, sh, sel=sel)
    assert torch.equal(ga3, ga_first) and torch.equal(gs3, gs2) and float(l3.sum()) == float(loss)
    with pytest.raises(RuntimeError, match="two forms"):
        hs.step(albedo, sh, idx=idx, sel=sel)
    with pytest.raises(RuntimeError, match="int32"):
        hs.step(albedo, sh, sel=sel.long())


def test_fused_stage1_step_vs_cpu_checker(oracle):
    from fmhr_amd import pipeline
    from fmhr_amd.stage1 import HamStage1
    dev = torch.device("cuda:0")
    sc = _tiny_capture(oracle)
    res = (sc["H"], sc["W"])
    torch.manual_seed(1)
    alb = (0.6 + 0.1 * torch.rand(1, sc["vertices"].shape[0], 3)).requires_grad_(True)
    shc = torch.zeros(sc["w2c"].shape[0], 9)
    shc[:, 0] = 0.8
    shc = (shc + 0.1 * torch.randn_like(shc)).requires_grad_(True)
    lc = pipeline.stage1_loss(oracle, None, sc["vertices"], sc["faces"], alb, shc, sc["w2c"], sc["proj"], sc["img"], sc["mask"], res, 50.0)
    lc.backward()
    g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "faces", "img", "mask", "w2c", "proj")}
    hs = HamStage1(g["vertices"], g["faces"], g["w2c"], g["proj"], g["img"], g["mask"], res, 50.0)
    lv, ga, gs = hs.step(alb.detach().to(dev).contiguous(), shc.detach().to(dev).contiguous())
    loss = lv.sum()
    # (the clip positions come from torch's einsum on each device: a handful of marginal pixels may fall on the other side of a rule,
    # as in test_gpu_pipeline.py::test_init_and_stage1_match_checker)
    assert abs(float(loss) - float(lc)) < 1e-3 * abs(float(lc))
    assert rel_err(ga, alb.grad[0]) < 2e-2 and rel_err(gs, shc.grad) < 2e-3


def test_fused_stage1_adam_matches_separate_optimizer_steps():
    from fmhr_amd.optim import Adam
    from fmhr_amd.stage1 import HamStage1
    dev = torch.device("cuda:0")
    sc, faces, w2c, proj, tg = _scene(dev, views=4, seed=2)
    vt = sc["vertices"].to(dev).contiguous()
    hs = HamStage1(vt, faces, w2c, proj, tg["img"], tg["mask"], (sc["H"], sc["W"]), 50.0)
    a0 = sc["albedo0"][None].to(dev).contiguous()
    s0 = sc["sh0"].to(dev).contiguous()
    runs = []
    for fused in (False, True):
        a = a0.clone().requires_grad_(True)
        s = s0.clone().requires_grad_(True)
        opt = Adam([{'params': a, 'lr': 5e-3}, {'params': s, 'lr': 5e-3}])
        for it in range(4):
            idx = torch.tensor([it % 4, (it + 2) % 4], device=dev)   # batches of two views: the other SH rows still move (momentum)
            if fused:
                hs.step(a, s, idx=idx, optimizer=opt, params=(a, s))
            else:
                _, ga, gs = hs.step(a, s, idx=idx)
                a.grad, s.grad = ga.view(1, -1, 3).clone(), gs.clone()
                opt.step()
        runs.append((a.detach().clone(), s.detach().clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("views", [8])
def test_fused_stage1_metric_mesh(views):
    """the metric mesh at 512 x 512: the fused step against the cached op-by-op form pipeline.run_ham used before"""
    from fmhr_amd import pipeline, synthetic
    from fmhr_amd.ham import render_forward
    from fmhr_amd.stage1 import HamStage1
    dev = torch.device("cuda:0")
    sc = synthetic.scene_from_config("metric", 0, views)
    H, W = sc["H"], sc["W"]
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].contiguous().to(dev), sc["proj"].contiguous().to(dev)
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    vt = sc["vertices"].to(dev).contiguous()
    ops = pipeline.product_ops()
    cache = pipeline.Stage1Cache(ops, ops.RasterizeGLContext(), vt, faces, w2c, proj, tg["mask"], (H, W))
    a = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
    s = sc["sh0"].to(dev).clone().requires_grad_(True)
    idx = torch.arange(views, device=dev)
    lk = pipeline.stage1_loss_cached(ops, cache, idx, a, s, tg["img"], 50.0)
    lk.backward()
    hs = HamStage1(vt, faces, w2c, proj, tg["img"], tg["mask"], (H, W), 50.0)
    lv, ga, gs = hs.step(a.detach(), s.detach())
    loss = lv.sum()
    assert abs(float(loss) - float(lk)) < TOL * abs(float(lk))
    assert rel_err(ga, a.grad[0]) < TOL and rel_err(gs, s.grad) < TOL


def test_fused_sh_initialisation_matches_the_per_view_lstsq_loop():
    """stage1.init_sh (normal equations accumulated on the GPU, 9 x 9 solves on the host) against pipeline.init_sh (the reference's loop
    :124-174: one view at a time, np.linalg.lstsq on the [Pv, 9] design matrix) on the product's generic ops"""
    from fmhr_amd import pipeline, stage1
    dev = torch.device("cuda:0")
    sc, faces, w2c, proj, tg = _scene(dev, views=5, seed=7)
    vt = sc["vertices"].to(dev).contiguous()
    res = (sc["H"], sc["W"])
    img = tg["img"]
    gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
    ref = pipeline.init_sh(pipeline.product_ops(), vt, faces, img, gray, tg["mask"], w2c, proj, res)
    hs = stage1.HamStage1(vt, faces, w2c, proj, img, tg["mask"], res, 50.0, chunk=2, want_pred_masks=True)
    got = stage1.init_sh(hs, gray)
    assert torch.equal(got["valid_masks"], ref["valid_masks"])
    assert rel_err(got["sh_coeffs"], ref["sh_coeffs"]) < 1e-4 and rel_err(got["sh_global"], ref["sh_global"]) < 1e-4
    assert rel_err(got["albedo_mean"], ref["albedo_mean"]) < 1e-5


def test_fused_stage1_edge_cases():
    """a view without a single valid pixel (empty mask), a batch that holds only that view, and a mesh that is off screen in a view"""
    from fmhr_amd import pipeline, stage1
    dev = torch.device("cuda:0")
    sc, faces, w2c, proj, tg = _scene(dev, views=4, seed=9)
    vt = sc["vertices"].to(dev).contiguous()
    res = (sc["H"], sc["W"])
    mask = tg["mask"].clone()
    mask[1] = 0.0                                                   # view 1: nothing valid
    w2c = w2c.clone()
    w2c[3, 3, 0] += 50.0                                            # view 3: the hand is far outside the frustum
    ops = pipeline.product_ops()
    albedo = sc["albedo0"][None].to(dev).contiguous()
    sh = sc["sh0"].to(dev).contiguous()
    hs = stage1.HamStage1(vt, faces, w2c, proj, tg["img"], mask, res, 50.0, want_pred_masks=True)
    vv = hs.view_valid.cpu()
    assert int(vv[1]) == 0 and int(vv[3]) == 0 and int(vv[0]) > 0
    lv, ga, gs = hs.step(albedo, sh)
    lr, gar, gsr = _reference(pipeline, ops, vt, faces, albedo, sh, w2c, proj, tg["img"], mask, res)
    assert abs(float(lv.sum()) - float(lr)) < TOL * abs(float(lr))
    assert rel_err(ga, gar) < TOL and rel_err(gs, gsr) < TOL
    assert float(gs[1].abs().max()) == 0.0 and float(gs[3].abs().max()) == 0.0 and float(lv[1]) == 0.0
    # a batch of empty views only: F.l1_loss over zero elements is NaN in torch; the fused step reports 0 and zero gradients
    lv, ga, gs = hs.step(albedo, sh, idx=torch.tensor([1, 3], device=dev))
    assert float(lv.abs().sum()) == 0.0 and float(ga.abs().max()) == 0.0 and float(gs.abs().max()) == 0.0
    # the initialisation copes with views that have no equations (zero coefficients for them)
    img = tg["img"]
    gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
    init = stage1.init_sh(hs, gray)
    assert float(init["sh_coeffs"][1].abs().max()) == 0.0 and torch.isfinite(init["sh_coeffs"]).all() and torch.isfinite(init["albedo_mean"]).all()


def test_sh_initialisation_of_a_rank_deficient_view_follows_the_reference():
    """a view with fewer valid pixels than SH coefficients (and one whose normals barely span the basis) must get the
    reference's minimum-norm least-squares solution of the [Pv, 9] DESIGN matrix (mesh_sfs_optim.py:153-154), not that of the squared
    normal equations; and HamStage1.step(optimizer=...) must refuse parameters that are not the storage behind its arguments."""
    from fmhr_amd import pipeline, stage1
    from fmhr_amd.optim import Adam
    dev = torch.device("cuda:0")
    sc, faces, w2c, proj, tg = _scene(dev, views=3, seed=11)
    vt = sc["vertices"].to(dev).contiguous()
    res = (sc["H"], sc["W"])
    img = tg["img"]
    gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
    mask = tg["mask"].clone()
    cov = torch.nonzero(mask[1] > 0)
    keep = cov[cov.shape[0] // 2: cov.shape[0] // 2 + 5]             # view 1: five valid pixels (< 9 unknowns)
    mask[1] = 0.0
    mask[1][keep[:, 0], keep[:, 1]] = 1.0
    ref = pipeline.init_sh(pipeline.product_ops(), vt, faces, img, gray, mask, w2c, proj, res)
    hs = stage1.HamStage1(vt, faces, w2c, proj, img, mask, res, 50.0, want_pred_masks=True)
    assert int(hs.view_valid[1]) == 5
    got = stage1.init_sh(hs, gray)
    # five neighbouring pixels have nearly the same normal: the minimum-norm coefficients are O(1e4) and move by ~5e-3 with the last
    # bit of the basis products (float32 products stored in a float64 matrix, as models/utils.py:188-204 forms them, vs float64
    # products in pipeline.init_sh); the squared normal equations would be off by orders of magnitude
    assert rel_err(got["sh_coeffs"][1], ref["sh_coeffs"][1]) < 2e-2, (got["sh_coeffs"][1], ref["sh_coeffs"][1])
    assert rel_err(got["sh_coeffs"][[0, 2]], ref["sh_coeffs"][[0, 2]]) < 1e-4 and rel_err(got["sh_global"], ref["sh_global"]) < 1e-4
    # the parameter check of the fused update
    a = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
    s = got["sh_coeffs"].clone().requires_grad_(True)
    opt = Adam([{'params': a, 'lr': 5e-3}, {'params': s, 'lr': 5e-3}])
    with pytest.raises(RuntimeError, match="params"):
        hs.step(a, s, optimizer=opt)
    with pytest.raises(RuntimeError, match="params"):
        hs.step(a.detach().clone(), s, optimizer=opt, params=(a, s))
    hs.step(a, s, optimizer=opt, params=(a, s))
    assert opt.state[a]['step'] == 1 and opt.state[s]['step'] == 1
