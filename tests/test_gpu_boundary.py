"""The drop-in boundary, exercised the way INTEGRATION.md advertises it: the reference's OWN statements of the SH initialisation
(/root/reference/mesh_sfs_optim.py:124-174) and of the stage-1 loop body (:196-240) executed on the import-swap surface

    import fmhr_amd.dr as dr
    from fmhr_amd.utils import get_normals, get_radiance, get_matrix, laplacian_smoothing

NumPy hand-offs (`get_matrix` on an np.ndarray, np.linalg.lstsq), boolean-mask gathers, `.cuda()` and torch.optim.Adam included,
next to tests/test_gpu_ham.py::test_unchanged_loop_body_on_dr, which does the same for the stage-2 body (:253-306).  The bodies
below restate the reference's lines as a test must ("the build's own loop must reproduce lines ... exactly"); the
results are compared with the same computation on the CPU checker."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.optim import Adam

from util import log_parity, mesh_scene, rel_err

pytestmark = pytest.mark.gpu


def _capture(oracle, dims=(4, 96, 80, 12, 16), seed=2):
    from test_gpu_ham import oracle_forward_fn
    from fmhr_amd import synthetic as S
    sc = mesh_scene(dims, seed=seed)
    with torch.no_grad():
        tg = S.make_targets(sc, oracle_forward_fn(oracle, sc))
    sc.update(tg)
    sc["mask"][:, : sc["H"] // 6] = 0                       # photometric mask != coverage somewhere
    sc["gray"] = 0.114 * sc["img"][..., 0] + 0.587 * sc["img"][..., 1] + 0.299 * sc["img"][..., 2]
    return sc


def _reference_init(vertices, faces, imgs, grayimgs, masks, w2cs, projs, resolution, degree):
    """mesh_sfs_optim.py:120-176 on the swap surface."""
    import fmhr_amd.dr as dr
    from fmhr_amd.utils import get_normals, get_radiance, get_matrix

    glctx = dr.RasterizeGLContext()

    num = imgs.shape[0]

    with torch.no_grad():
        valid_normals = []
        valid_grayimgs = []
        valid_masks = []
        valid_imgs = []
        sh_coeffs = []
        for k in range(0, num, 1):
            n = min(num, k + 1) - k
            w2c = w2cs[k:k + 1]
            proj = projs[k:k + 1]
            mask = masks[k:k + 1]
            img = imgs[k:k + 1]
            grayimg = grayimgs[k:k + 1]

            vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], axis=1).unsqueeze(0).expand(n, -1, -1)
            rot_verts = torch.einsum('ijk,ikl->ijl', vertsw, w2c)
            proj_verts = torch.einsum('ijk,ikl->ijl', rot_verts, proj)
            normals = get_normals(vertsw[:, :, :3], faces.long())
            rast_out, _ = dr.rasterize(glctx, proj_verts, faces, resolution=resolution)
            feat, _ = dr.interpolate(torch.cat([normals, torch.ones_like(vertsw[:, :, :1])], 2), rast_out, faces)
            pred_normals = feat[:, :, :, :3].contiguous()
            pred_mask = feat[:, :, :, 3:4].contiguous()
            pred_mask = dr.antialias(pred_mask, rast_out, proj_verts, faces).squeeze(-1)
            pred_normals = dr.antialias(pred_normals, rast_out, proj_verts, faces)
            pred_normals = F.normalize(pred_normals, p=2, dim=3)

            valid_idx = (mask > 0) & (rast_out[:, :, :, 3] > 0)
            valid_normal = pred_normals[valid_idx].detach().cpu().numpy()
            valid_grayimg = grayimg[valid_idx].detach().cpu().numpy()
            matrix = get_matrix(valid_normal, degree)
            sh_coeff = np.linalg.lstsq(matrix, valid_grayimg, rcond=None)[0]
            valid_normals.append(valid_normal)
            valid_imgs.append(img[valid_idx])
            valid_grayimgs.append(grayimg[valid_idx].detach().cpu().numpy())
            valid_masks.append(pred_mask)
            sh_coeffs.append(torch.from_numpy(sh_coeff.astype(np.float32)).cuda().unsqueeze(0))
    valid_normals = np.concatenate(valid_normals, axis=0)
    valid_imgs = torch.cat(valid_imgs, 0)
    valid_grayimgs = np.concatenate(valid_grayimgs, axis=0)
    valid_masks = torch.cat(valid_masks, 0)

    matrix = get_matrix(valid_normals, degree)
    assert isinstance(matrix, np.ndarray) and matrix.dtype == np.float64 and matrix.shape[1] == degree ** 2
    sh_coeff = np.linalg.lstsq(matrix, valid_grayimgs, rcond=None)[0]

    sh_coeff = torch.from_numpy(sh_coeff.astype(np.float32)).cuda()
    sh_coeffs = torch.cat(sh_coeffs, 0)
    sh_coeffs.requires_grad_(True)

    radiance = get_radiance(sh_coeff, torch.from_numpy(valid_normals).cuda(), degree).unsqueeze(-1)
    albedo_mean = (valid_imgs / radiance).mean(0).detach()

    albedo = (torch.zeros_like(vertices) + albedo_mean).unsqueeze(0)
    albedo.requires_grad_(True)
    return dict(glctx=glctx, sh_coeffs=sh_coeffs, sh_global=sh_coeff, albedo_mean=albedo_mean, albedo=albedo, valid_masks=valid_masks)


def test_reference_init_statements_on_the_swap_surface(oracle):
    from fmhr_amd import pipeline
    sc = _capture(oracle)
    dev = torch.device("cuda:0")
    res = (sc["H"], sc["W"])
    g = {k: sc[k].to(dev) for k in ("vertices", "faces", "img", "gray", "mask", "w2c", "proj")}
    out = _reference_init(g["vertices"], g["faces"], g["img"], g["gray"], g["mask"], g["w2c"], g["proj"], res, 3)
    ref = pipeline.init_sh(oracle, sc["vertices"], sc["faces"], sc["img"], sc["gray"], sc["mask"], sc["w2c"], sc["proj"], res)
    e = dict(sh=rel_err(out["sh_coeffs"], ref["sh_coeffs"]), shg=rel_err(out["sh_global"], ref["sh_global"]),
             am=rel_err(out["albedo_mean"], ref["albedo_mean"]), vm=rel_err(out["valid_masks"], ref["valid_masks"]))
    log_parity("reference init statements (:124-174) on fmhr_amd.dr / fmhr_amd.utils vs the CPU checker: per-view SH %(sh).1e, global SH "
               "%(shg).1e, albedo_mean %(am).1e, start silhouettes %(vm).1e" % e)
    assert e["sh"] < 1e-4 and e["shg"] < 1e-4 and e["am"] < 1e-4 and e["vm"] < 1e-5, e
    assert out["albedo"].shape == (1, sc["V"], 3) and out["albedo"].requires_grad
    # degrees 1 and 2 of the same statements (conf key `degree`): get_matrix / get_radiance take them like the reference's do
    for degree in (1, 2):
        o = _reference_init(g["vertices"], g["faces"], g["img"], g["gray"], g["mask"], g["w2c"], g["proj"], res, degree)
        assert o["sh_coeffs"].shape == (sc["views"], degree ** 2) and torch.isfinite(o["albedo_mean"]).all()


def test_reference_stage1_statements_on_the_swap_surface(oracle):
    """mesh_sfs_optim.py:189-240, two epochs of it, on the swap surface with torch.optim.Adam; the first batch's loss and gradients
    against the same batch on the CPU checker (pipeline.stage1_loss on the oracle ops)."""
    import fmhr_amd.dr as dr
    from fmhr_amd import pipeline
    from fmhr_amd.utils import get_normals, get_radiance, laplacian_smoothing
    sc = _capture(oracle)
    dev = torch.device("cuda:0")
    resolution = (sc["H"], sc["W"])
    g = {k: sc[k].to(dev) for k in ("vertices", "faces", "img", "gray", "mask", "w2c", "proj")}
    vertices, faces, imgs, masks, w2cs, projs = g["vertices"], g["faces"], g["img"], g["mask"], g["w2c"], g["proj"]
    init = _reference_init(vertices, faces, imgs, g["gray"], masks, w2cs, projs, resolution, 3)
    glctx, sh_coeffs, albedo = init["glctx"], init["sh_coeffs"], init["albedo"]
    num, batch, degree = imgs.shape[0], 3, 3
    sfs_weight, albedo_weight, albedo_lr, sh_lr, epoch_albedo = 50.0, 1.0, 5e-3, 5e-3, 2
    delta = torch.zeros_like(vertices)
    delta.requires_grad_(True)
    vertices_tmp = torch.clone(vertices).detach()
    albedo0, sh0 = albedo.detach().clone(), sh_coeffs.detach().clone()

    vertices = (vertices_tmp + delta).detach()
    optimizer = Adam([{'params': albedo, 'lr': albedo_lr}, {'params': sh_coeffs, 'lr': sh_lr}])
    delta.requires_grad_(False)
    first, losses = None, []
    torch.manual_seed(0)
    for i in range(epoch_albedo):
        perm = torch.randperm(num).cuda()
        for k in range(0, num, batch):
            n = min(num, k + batch) - k
            w2c = w2cs[perm[k:k + batch]]
            proj = projs[perm[k:k + batch]]
            img = imgs[perm[k:k + batch]]
            mask = masks[perm[k:k + batch]]
            sh_coeff = sh_coeffs[perm[k:k + batch]]

            vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], axis=1).unsqueeze(0).expand(n, -1, -1)
            rot_verts = torch.einsum('ijk,ikl->ijl', vertsw, w2c)
            proj_verts = torch.einsum('ijk,ikl->ijl', rot_verts, proj)
            normals = get_normals(vertsw[:, :, :3], faces.long())

            rast_out, _ = dr.rasterize(glctx, proj_verts, faces, resolution=resolution)
            feat = torch.cat([normals, albedo.expand(n, -1, -1)], dim=2)
            feat, _ = dr.interpolate(feat, rast_out, faces)
            pred_normals = feat[:, :, :, :3].contiguous()
            rast_albedo = feat[:, :, :, 3:6].contiguous()
            pred_normals = dr.antialias(pred_normals, rast_out, proj_verts, faces)
            pred_normals = F.normalize(pred_normals, p=2, dim=3)
            rast_albedo = dr.antialias(rast_albedo, rast_out, proj_verts, faces)

            valid_idx = torch.where((mask > 0) & (rast_out[:, :, :, 3] > 0))
            valid_normals = pred_normals[valid_idx]
            valid_shcoeff = sh_coeff[valid_idx[0]]
            valid_albedo = rast_albedo[valid_idx]

            valid_img = img[valid_idx]
            radiance = get_radiance(valid_shcoeff, valid_normals, degree).unsqueeze(-1)
            pred_img = radiance * valid_albedo

            sfs_loss = sfs_weight * F.l1_loss(pred_img, valid_img)
            albedo_loss = albedo_weight * laplacian_smoothing(albedo.squeeze(0), faces.long(), method="uniform")

            loss = sfs_loss

            optimizer.zero_grad()
            loss.backward()
            if first is None:
                first = dict(idx=perm[k:k + batch].cpu(), loss=float(loss), galbedo=albedo.grad.clone(), gsh=sh_coeffs.grad.clone(),
                             albedo_loss=float(albedo_loss))
            optimizer.step()
            losses.append(sfs_loss.item())

    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # the first batch on the CPU checker
    idx = first["idx"]
    a = albedo0.cpu().double().clone().requires_grad_(True)
    s = sh0.cpu().double().clone().requires_grad_(True)
    d = lambda t: t.double()
    with oracle.f32_decisions():
        ref = pipeline.stage1_loss(oracle, None, d(sc["vertices"]), sc["faces"], a, s[idx], d(sc["w2c"][idx]), d(sc["proj"][idx]),
                                   d(sc["img"][idx]), d(sc["mask"][idx]), resolution, sfs_weight)
        ref.backward()
    gsh_ref = torch.zeros_like(s)
    gsh_ref = s.grad
    e = (abs(first["loss"] - float(ref)) / abs(float(ref)), rel_err(first["galbedo"], a.grad), rel_err(first["gsh"], gsh_ref))
    log_parity("reference stage-1 statements (:196-240) on the swap surface vs the CPU checker (fp64 on fp32 decisions): loss %.1e, "
               "albedo.grad %.1e, sh_coeffs.grad %.1e" % e)
    assert e[0] < 1e-4 and e[1] < 1e-4 and e[2] < 1e-4, e
    assert first["albedo_loss"] >= 0.0
