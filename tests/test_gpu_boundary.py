"""The drop-in boundary against a fixture made from the reference's OWN statements.

tests/golden/boundary_loop.npz holds inputs and outputs of lines 120-317 of /root/reference/mesh_sfs_optim.py — SH initialisation,
stage 1, stage 2 — EXECUTED in the build container over the CPU checker's nvdiffrast-shaped ops and the reference's own helpers
(tests/golden/make_boundary_goldens.py reads them from the reference at run time; the fixture is arrays only).  Here the PRODUCT runs the
same computation through the build's own statements on the import-swap surface INTEGRATION.md advertises

    import fmhr_amd.dr as dr
    from fmhr_amd.utils import get_normals, get_radiance, get_matrix, laplacian_smoothing

with the reference's call shapes — NumPy hand-offs (`get_matrix` on an np.ndarray, np.linalg.lstsq), `[1,V,C]` / `[n,V,C]` attributes,
boolean-mask gathers, torch.optim.Adam, a fresh permutation and a ragged last batch per epoch — and is compared with the arrays:
fmhr_amd.pipeline.init_sh / stage1_loss on fmhr_amd.dr + fmhr_amd.utils, and the fused HamStage2 step for stage 2.  (The stage-2 body
on the op-by-op surface: tests/test_gpu_ham.py::test_unchanged_loop_body_on_dr.)"""
import os

import numpy as np
import pytest
import torch
from torch.optim import Adam

from util import log_parity, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "boundary_loop.npz"))
    conf = dict(zip([str(k) for k in g["conf_keys"]], [float(v) for v in g["conf_values"]]))
    return g, conf


def _inputs(g, dev):
    t = lambda k: torch.from_numpy(g["in_" + k]).to(dev)
    return dict(vertices=t("vertices"), faces=t("faces"), img=t("img"), gray=t("gray"), mask=t("mask"), w2c=t("w2c"), proj=t("proj"))


def test_init_on_the_swap_surface_matches_the_reference_statements(fx):
    """mesh_sfs_optim.py:120-177 (per-view + global SH least squares, albedo_mean, antialiased start silhouettes)"""
    from fmhr_amd import pipeline
    g, conf = fx
    dev = torch.device("cuda:0")
    x = _inputs(g, dev)
    res = tuple(g["in_img"].shape[1:3])
    ops = pipeline.product_ops()
    out = pipeline.init_sh(ops, x["vertices"], x["faces"], x["img"], x["gray"], x["mask"], x["w2c"], x["proj"], res, 3)
    e = dict(sh=rel_err(out["sh_coeffs"], g["init_sh_coeffs"]), shg=rel_err(out["sh_global"], g["init_sh_global"]),
             am=rel_err(out["albedo_mean"], g["init_albedo_mean"]), vm=rel_err(out["valid_masks"], g["init_valid_masks"]))
    log_parity("init (:120-177): the product's ops under the build's statements vs the reference's statements executed over the CPU checker: "
               "per-view SH %(sh).1e, global SH %(shg).1e, albedo_mean %(am).1e, start silhouettes %(vm).1e" % e)
    assert e["sh"] < 1e-4 and e["shg"] < 1e-4 and e["am"] < 1e-4 and e["vm"] < 1e-5, e
    # get_matrix is handed NumPy and hands NumPy back, in float64, like models/utils.py:188-190
    m = ops.get_matrix(np.array([[0.6, 0.0, 0.8], [0.0, 1.0, 0.0]]), 3)
    assert isinstance(m, np.ndarray) and m.dtype == np.float64 and m.shape == (2, 9)
    # degrees 1 and 2 of the same statements (conf key `degree`)
    for degree in (1, 2):
        o = pipeline.init_sh(ops, x["vertices"], x["faces"], x["img"], x["gray"], x["mask"], x["w2c"], x["proj"], res, degree)
        assert o["sh_coeffs"].shape == (g["in_img"].shape[0], degree ** 2)
        ed = (rel_err(o["sh_coeffs"], g["init_sh_coeffs_deg%d" % degree]), rel_err(o["sh_global"], g["init_sh_global_deg%d" % degree]),
              rel_err(o["albedo_mean"], g["init_albedo_mean_deg%d" % degree]))
        assert max(ed) < 1e-4, (degree, ed)
    # the fused form the product's run_ham uses (HamStage1 + normal equations on the GPU) gives the same initialisation
    from fmhr_amd import stage1 as s1
    hs1 = s1.HamStage1(x["vertices"], x["faces"], x["w2c"], x["proj"], x["img"], x["mask"], res, conf["sfs_weight"], want_pred_masks=True)
    fi = s1.init_sh(hs1, x["gray"])
    ef = (rel_err(fi["sh_coeffs"], g["init_sh_coeffs"]), rel_err(fi["sh_global"], g["init_sh_global"]), rel_err(fi["albedo_mean"], g["init_albedo_mean"]),
          rel_err(fi["valid_masks"], g["init_valid_masks"]))
    log_parity("init, fused form (HamStage1 + GPU normal equations) vs the same fixture: per-view SH %.1e, global SH %.1e, albedo_mean %.1e, start silhouettes %.1e" % ef)
    assert max(ef[:3]) < 2e-4 and ef[3] < 1e-5, ef
    for degree in (1, 2):                                              # round 6: degrees 1 and 2 on the same accumulated normal equations (their leading blocks)
        fd = s1.init_sh(hs1, x["gray"], degree=degree)
        assert fd["sh_coeffs"].shape == (g["in_img"].shape[0], degree ** 2)
        ed = (rel_err(fd["sh_coeffs"], g["init_sh_coeffs_deg%d" % degree]), rel_err(fd["sh_global"], g["init_sh_global_deg%d" % degree]),
              rel_err(fd["albedo_mean"], g["init_albedo_mean_deg%d" % degree]))
        log_parity("init, fused form, degree %d vs the fixture's reference statements: per-view SH %.1e, global SH %.1e, albedo_mean %.1e" % ((degree,) + ed))
        assert max(ed) < 2e-4, (degree, ed)


def test_stage1_on_the_swap_surface_matches_the_reference_statements(fx):
    """mesh_sfs_optim.py:179-240: two epochs, batches of 3 of 4 views in the fixture's permutations, torch.optim.Adam — every batch's
    loss, the first batch's gradients and the parameters after the last step against the fixture (fp64 values on the fp32 decisions)"""
    from fmhr_amd import pipeline
    g, conf = fx
    dev = torch.device("cuda:0")
    x = _inputs(g, dev)
    res = tuple(g["in_img"].shape[1:3])
    ops = pipeline.product_ops()
    glctx = ops.RasterizeGLContext()
    num, batch = x["img"].shape[0], int(conf["batch"])
    # the fixture's own initialisation: the comparison of stage 1 does not inherit the initialisation's differences
    sh_coeffs = torch.from_numpy(g["init_sh_coeffs"]).to(dev).clone().requires_grad_(True)
    albedo = (torch.zeros_like(x["vertices"]) + torch.from_numpy(g["init_albedo_mean"]).to(dev)).unsqueeze(0).contiguous().requires_grad_(True)
    optimizer = Adam([{'params': albedo, 'lr': conf["albedo_lr"]}, {'params': sh_coeffs, 'lr': conf["sh_lr"]}])
    losses, first = [], None
    for perm in torch.from_numpy(g["s1_perms"]).to(dev):
        for k in range(0, num, batch):
            idx = perm[k:k + batch]
            loss = pipeline.stage1_loss(ops, glctx, x["vertices"], x["faces"], albedo, sh_coeffs[idx], x["w2c"][idx], x["proj"][idx],
                                        x["img"][idx], x["mask"][idx], res, conf["sfs_weight"])
            optimizer.zero_grad()
            loss.backward()
            if first is None:
                first = (albedo.grad.clone(), sh_coeffs.grad.clone())
            optimizer.step()
            losses.append(float(loss))
    ref_losses = conf["sfs_weight"] * g["s1_l1"]
    assert len(losses) == int(g["s1_steps"]) == len(ref_losses)
    e = (max(abs(a - b) / abs(b) for a, b in zip(losses, ref_losses)), rel_err(first[0], g["s1_first_galbedo"]), rel_err(first[1], g["s1_first_gsh"]),
         rel_err(albedo.detach(), g["s1_albedo"]), rel_err(sh_coeffs.detach(), g["s1_sh_coeffs"]))
    log_parity("stage 1 (:179-240), %d Adam steps in the fixture's permutations: losses %.1e, first batch albedo.grad %.1e sh_coeffs.grad %.1e, "
               "final albedo %.1e sh_coeffs %.1e" % ((len(losses),) + e))
    assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4 and e[3] < 1e-4 and e[4] < 1e-4, e


def test_stage2_fused_step_matches_the_reference_statements(fx):
    """mesh_sfs_optim.py:242-317, two epochs in the fixture's permutations (a batch of 3 views, then the last view alone; albedo_weight
    divided by 10000 from the second epoch on, :250-251): the fused
    HamStage2 step + fmhr_amd.optim.Adam as run_ham drives them — the first batch's L1 term and gradients, the last batch's six loss
    terms, the parameters after the epoch — against what the reference's statements produced."""
    from fmhr_amd import optim
    from fmhr_amd.ham import HamStage2
    g, conf = fx
    dev = torch.device("cuda:0")
    x = _inputs(g, dev)
    H, W = g["in_img"].shape[1:3]
    V, num, batch = x["vertices"].shape[0], x["img"].shape[0], int(conf["batch"])
    weights = dict(sfs=conf["sfs_weight"], lap=conf["lap_weight"], albedo=conf["albedo_weight"], mask=conf["mask_weight"],
                   edge=conf["edge_weight"], delta=conf["delta_weight"])
    elm = float(g["init_edge_length_mean"])
    sh = torch.from_numpy(g["s1_sh_coeffs"]).to(dev).float().contiguous()               # the end state of the fixture's stage 1
    albedo = torch.from_numpy(g["s1_albedo"]).to(dev).float().contiguous().requires_grad_(True)
    valid_masks = torch.from_numpy(g["init_valid_masks"]).to(dev)
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    opt = optim.Adam([{'params': delta, 'lr': conf["lr"]}, {'params': albedo, 'lr': conf["albedo_lr"]}])
    epoch_sfs = int(conf["epoch_sfs"])
    first, lds, steppers = None, [], {}
    for ep, perm in enumerate(torch.from_numpy(g["s2_perms"]).to(dev)):
        if ep == epoch_sfs // 2:                                                         # :250-251, as run_ham applies it
            weights["albedo"] = weights["albedo"] / 10000
            steppers.clear()
        for k in range(0, num, batch):
            idx = perm[k:k + batch]
            n = idx.shape[0]
            if n not in steppers:
                steppers[n] = HamStage2(x["faces"], V, n, H, W, weights, elm)
            hs = steppers[n]
            losses, gd, ga = hs.step(x["vertices"], delta.detach(), albedo.detach()[0], sh[idx].contiguous(), x["w2c"][idx].contiguous(),
                                     x["proj"][idx].contiguous(), x["img"][idx].contiguous(), x["mask"][idx].contiguous(), valid_masks[idx].contiguous())
            lds.append(hs.loss_dict())
            if first is None:
                first = (gd.clone(), ga.clone())
            delta.grad, albedo.grad = gd, ga.view(1, V, 3)
            opt.step()
    assert len(lds) == int(g["s2_steps"])
    ref_last = dict(zip(("sfs", "lap", "albedo", "mask", "edge", "delta", "total"), g["s2_last_losses"]))
    e_l1 = [abs(ld["sfs"] - conf["sfs_weight"] * r) / abs(conf["sfs_weight"] * r) for ld, r in zip(lds, g["s2_l1"])]
    e_last = max(abs(lds[-1][k] - v) / max(abs(v), 1e-6) for k, v in ref_last.items())
    e = (max(e_l1), e_last, rel_err(first[0], g["s2_first_gdelta"]), rel_err(first[1], g["s2_first_galbedo"][0]),
         rel_err(delta.detach(), g["s2_delta"]), rel_err(albedo.detach(), g["s2_albedo"]))
    log_parity("stage 2 (:242-317), fused HamStage2 step + fmhr Adam, two epochs of (3 views + 1 view): L1 terms %.1e, the last batch's six losses %.1e, "
               "first batch delta.grad %.1e albedo.grad %.1e, final delta %.1e albedo %.1e" % e)
    assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4 and e[3] < 1e-4, e
    # four Adam steps from zero moments are ~lr x sign(g) each: a component whose gradient is rounding noise may differ by O(lr)
    assert e[4] < 1e-3 and e[5] < 1e-4, e                                                 # (measured 6.5e-5 / 2.3e-7)


def test_stage2_loop_of_the_product_matches_the_reference_statements(fx):
    """the same fixture through the loop run_ham DRIVES (fmhr_amd.pipeline.Stage2Loop: the capture resident, the batch's views selected
    inside the step — ABI 6 —, the fused Adam, the albedo-weight switch of :250-251 applied to the live workspaces), fed the fixture's
    permutations instead of its own draws: the last batch's six loss terms and the parameters after the two epochs"""
    from fmhr_amd import optim, pipeline
    g, conf = fx
    dev = torch.device("cuda:0")
    x = _inputs(g, dev)
    H, W = g["in_img"].shape[1:3]
    V, batch = x["vertices"].shape[0], int(conf["batch"])
    weights = dict(sfs=conf["sfs_weight"], lap=conf["lap_weight"], albedo=conf["albedo_weight"], mask=conf["mask_weight"],
                   edge=conf["edge_weight"], delta=conf["delta_weight"])
    sh = torch.from_numpy(g["s1_sh_coeffs"]).to(dev).float().contiguous()
    albedo = torch.from_numpy(g["s1_albedo"]).to(dev).float().contiguous().requires_grad_(True)
    valid_masks = torch.from_numpy(g["init_valid_masks"]).to(dev).contiguous()
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    opt = optim.Adam([{'params': delta, 'lr': conf["lr"]}, {'params': albedo, 'lr': conf["albedo_lr"]}])
    perms = iter(torch.from_numpy(g["s2_perms"]).to(torch.int32).to(dev))

    class FixturePerms(pipeline.Stage2Loop):
        def _next_perm(self):
            return next(perms)
    loop = FixturePerms(x["vertices"], x["faces"], delta, albedo, sh, x["img"].contiguous(), x["mask"].contiguous(), valid_masks, x["w2c"].contiguous(),
                        x["proj"].contiguous(), (H, W), weights, float(g["init_edge_length_mean"]), opt, batch, int(conf["epoch_sfs"]),
                        torch.Generator().manual_seed(0))
    for _ in range(g["s2_perms"].shape[0]):
        loop.run_epoch()
    loop.check()
    assert len(loop.steppers) == 2                                      # a batch of 3 views and the ragged last one: both workspaces lived through the switch
    ref_last = dict(zip(("sfs", "lap", "albedo", "mask", "edge", "delta", "total"), g["s2_last_losses"]))
    ld = loop.hs.loss_dict()
    e_last = max(abs(ld[k] - v) / max(abs(v), 1e-6) for k, v in ref_last.items())
    e = (e_last, rel_err(delta.detach(), g["s2_delta"]), rel_err(albedo.detach(), g["s2_albedo"]))
    log_parity("stage 2 (:242-317) through the product's own loop (Stage2Loop, views selected inside the step, fused Adam, albedo-weight switch on the "
               "live workspaces), the fixture's permutations: the last batch's six losses %.1e, final delta %.1e albedo %.1e" % e)
    assert e[0] < 1e-4 and e[1] < 1e-3 and e[2] < 1e-4, e
