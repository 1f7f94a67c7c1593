"""oracle/checker_scenes.py — TEST INFRASTRUCTURE, NOT PRODUCT: the seeded scenes the parity checks run on (synthetic mesh + cameras from
fmhr_amd.synthetic, targets rendered BY THE CPU ORACLE) and the hand-over of the loop body's discrete decisions from the device under test to
the fp64 reference.  Used by tests/ and by __graft_entry__.smoke(); nothing under fmhr_amd/ imports it."""
import numpy as np
import torch


def mesh_scene(name_or_dims, seed=0, views=None):
    """synthetic HAM scene (fmhr_amd.synthetic) as CPU tensors + clip positions for its start mesh."""
    from fmhr_amd import synthetic as S
    if isinstance(name_or_dims, str):
        sc = S.scene_from_config(name_or_dims, seed, views)
    else:                                            # (views, H, W, rings, segs[, subdiv[, hands]])
        d = tuple(name_or_dims)
        sc = S.make_scene(*d[:6], seed=seed, hands=d[6] if len(d) > 6 else 1)
    v4 = torch.cat([sc["vertices"], torch.ones(sc["V"], 1)], 1)
    sc["pos"] = torch.einsum('nvk,nkl->nvl', torch.einsum('vk,nkl->nvl', v4, sc["w2c"]), sc["proj"]).contiguous()
    return sc


def oracle_forward_fn(oracle, sc):
    def fwd(vertices, albedo, sh, mask):
        z = torch.zeros_like(vertices)
        img = torch.zeros(sc["views"], sc["H"], sc["W"], 3)
        out = oracle.ham_stage2_losses(vertices, z, albedo[None], sh, sc["w2c"], sc["proj"], img, mask, mask, sc["faces"],
                                       (sc["H"], sc["W"]), dict(sfs=0, lap=0, albedo=0, mask=0, edge=0, delta=0), 0.0)
        return dict(tmp_img=out["tmp_img"], pred_mask=out["pred_mask"], rast=out["rast"])
    return fwd


def prepare(oracle, dims, seed):
    from fmhr_amd import synthetic as S
    sc = mesh_scene(dims, seed=seed)
    with torch.no_grad():
        tg = S.make_targets(sc, oracle_forward_fn(oracle, sc))
    rng = np.random.default_rng(seed + 100)
    sc["delta"] = torch.tensor(rng.standard_normal((sc["V"], 3)) * 2e-4, dtype=torch.float32)
    sc["albedo"] = (sc["albedo0"] + torch.tensor(rng.uniform(-0.1, 0.1, (sc["V"], 3)), dtype=torch.float32))
    # make the photometric mask differ from coverage somewhere (exercise mask>0 gating)
    tg["mask"][:, : sc["H"] // 5] = 0
    sc.update(tg)
    sc["elm"] = float(oracle.edge_length_mean(sc["vertices"], sc["faces"]))
    return sc


def gpu_loop_decisions(ref32, tmp_img_gpu, img, mask, name="", log=None):
    """The loop body's own discrete decisions as the GPU took them.  The rasterizer's decisions are bit-identical between
    the kernels and the fp32 twin (asserted: rast, antialias items), so the fp64 reference can take them from the twin; the SIGN of an
    L1 residual (mesh_sfs_optim.py:289) is decided on image VALUES, which agree to 1e-6 only — a residual within that of zero may
    come out with the other sign on the GPU (one pixel-channel in 2.6 M on the demo-capture scene: +-w_sfs / (3 Pv) x radiance at the
    three corners of one face = 2.5 % of that case's largest albedo gradient).  "fp64 values on the GPU's decisions" therefore takes
    the signs from the GPU's own antialiased image — after checking that they differ from the twin's ONLY where both residuals are
    within fp32 rounding of zero, and only in a handful of places: anything else is a defect, not a marginal decision."""
    idx = torch.where((mask > 0) & (ref32["rast"][..., 3] > 0))
    r_gpu = torch.as_tensor(tmp_img_gpu).detach().cpu().to(torch.float32)[idx] - img[idx]
    r_ref = ref32["tmp_img"].detach().to(torch.float32)[idx] - img[idx]
    s_gpu, s_ref = torch.sign(r_gpu), ref32["decisions"]["l1_sign"].to(torch.float32)
    assert s_gpu.shape == s_ref.shape
    flip = s_gpu != s_ref
    nflip = int(flip.sum())
    if nflip:
        tol = 4e-6 * max(1.0, float(img.abs().max()))
        assert nflip <= 8 and float(r_gpu[flip].abs().max()) <= tol and float(r_ref[flip].abs().max()) <= tol, \
            (name, nflip, float(r_gpu[flip].abs().max()), float(r_ref[flip].abs().max()))
        (log or (lambda line: None))(f"{name}: {nflip} L1 residual sign(s) within fp32 rounding of zero decided the other way by the GPU "
                   f"(|d| <= {float(torch.maximum(r_gpu[flip].abs(), r_ref[flip].abs()).max()):.1e}): the fp64 reference takes the GPU's")
    return dict(ref32["decisions"], l1_sign=s_gpu)
