"""fmhr_amd.examples.loop_body — what the two-line import swap of INTEGRATION.md looks like from the caller's side: the stage-2 loop body of
/root/reference/mesh_sfs_optim.py:253-306, restated statement by statement, on the product's nvdiffrast-shaped surface (fmhr_amd.dr) and
the models/utils.py-shaped helpers (fmhr_amd.utils) with torch autograd — op by op, nothing fused.  bench.py times it (`import_swap`),
the parity tests compare it with the fused step and with the CPU checker (tests/test_gpu_ham.py::test_unchanged_loop_body_on_dr).
It is an example and a measurement subject, not the product's loop: run_ham drives the fused HamStage2 step (fmhr_amd.pipeline)."""
import torch
import torch.nn.functional as F


def loop_body_on_dr(g, faces, n, H, W, weights, elm, pos_value=None, reg_dtype=torch.float32, lap_fn=None):
    """the reference's own loop body (mesh_sfs_optim.py:253-306) on fmhr_amd.dr / fmhr_amd.utils GPU ops.  `pos_value`
    substitutes the VALUE of the clip positions (straight-through: the gradient still flows through the einsums), so two
    implementations can be compared on identical discrete decisions.  `reg_dtype=float64` evaluates the mesh regularisers
    (torch ops only) in double on the fp32-rounded vertices: on a sub3-class mesh the uniform Laplacian cancels 3 digits
    (|Lx| ~ 1e-4 |x|), so two fp32 evaluations of it differ by more than 1e-4 and neither can referee the other."""
    from fmhr_amd import dr, utils
    delta = g["delta"].clone().requires_grad_(True)
    albedo = g["albedo"][None].clone().requires_grad_(True)
    glctx = dr.RasterizeGLContext()
    vertices = g["vertices"] + delta
    vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], axis=1).unsqueeze(0).expand(n, -1, -1)
    rot_verts = torch.einsum('ijk,ikl->ijl', vertsw, g["w2c"])
    proj_verts = torch.einsum('ijk,ikl->ijl', rot_verts, g["proj"])
    if pos_value is not None:
        proj_verts = pos_value.detach() + (proj_verts - proj_verts.detach())
    normals = utils.get_normals(vertsw[:, :, :3], faces.long())
    rast_out, _ = dr.rasterize(glctx, proj_verts, faces, resolution=(H, W))
    feat = torch.cat([normals, albedo.expand(n, -1, -1), torch.ones_like(vertsw[:, :, :1])], dim=2)
    feat, _ = dr.interpolate(feat, rast_out, faces)
    pred_normals = F.normalize(feat[:, :, :, :3].contiguous(), p=2, dim=3)
    rast_albedo = feat[:, :, :, 3:6].contiguous()
    pred_mask = dr.antialias(feat[:, :, :, 6:7].contiguous(), rast_out, proj_verts, faces).squeeze(-1)
    valid_idx = torch.where((g["mask"] > 0) & (rast_out[:, :, :, 3] > 0))
    radiance = utils.get_radiance(g["sh0"][valid_idx[0]], pred_normals[valid_idx], 3).unsqueeze(-1)
    pred_img = radiance * rast_albedo[valid_idx]
    tmp_img = torch.zeros_like(g["img"])
    tmp_img[valid_idx] = pred_img
    tmp_img = dr.antialias(tmp_img, rast_out, proj_verts, faces)
    terms = {}
    terms["sfs"] = weights["sfs"] * F.l1_loss(tmp_img[valid_idx], g["img"][valid_idx])
    rv, ra, rd = vertices.to(reg_dtype), albedo.squeeze(0).to(reg_dtype), delta.to(reg_dtype)
    fl = faces.long()
    if lap_fn is None:
        lap_fn = utils.laplacian_smoothing
    else:                                             # the checker's torch Laplacian builds its operator on the CPU
        rv, ra, rd, fl = rv.cpu(), ra.cpu(), rd.cpu(), fl.cpu()
    terms["lap"] = (weights["lap"] * lap_fn(rv, fl, method="uniform")).to(delta.device)
    terms["albedo"] = (weights["albedo"] * lap_fn(ra, fl, method="uniform")).to(delta.device)
    terms["mask"] = weights["mask"] * F.mse_loss(pred_mask, g["valid_mask"])
    a, b, c = rv[fl[:, 0]], rv[fl[:, 1]], rv[fl[:, 2]]
    edge_length = torch.cat([((a - b) ** 2).sum(1), ((c - b) ** 2).sum(1), ((a - c) ** 2).sum(1)])
    terms["edge"] = (torch.clip(edge_length - float(elm), 0, 1).mean() * weights["edge"]).to(delta.device)
    terms["delta"] = ((rd ** 2).sum(1).mean() * weights["delta"]).to(delta.device)
    loss = terms["sfs"] + terms["lap"] + terms["albedo"] + terms["mask"] + terms["delta"] + terms["edge"]
    loss.backward()
    return dict(loss=loss.detach(), terms={k: float(v) for k, v in terms.items()}, delta=delta, albedo=albedo,
                rast=rast_out.detach(), pos=proj_verts.detach(), tmp_img=tmp_img.detach(), n_valid=valid_idx[0].numel())
