"""fmhr_amd.stage1 — the fused stage-1 ("albedo") iteration of /root/reference/mesh_sfs_optim.py:196-240.

Stage 1 optimises the albedo [1,V,3] and the SH coefficients [num,9] with the vertices frozen (:191, :195), so everything geometric
of its loop body (:205-219) — rasterization, interpolated + antialiased + normalised normals, the validity mask, and the antialias
operator itself, a linear map of its colour input once the geometry is fixed — is the same in every iteration.  `HamStage1` computes
that constant part ONCE with the generic ops (fmhr_amd.dr / fmhr_amd.utils: the same C-ABI kernels `stage1_loss` uses), hands it to
the library's plan builder (include/fmhr_ham.h: fmhr_stage1_rows / _row_ptr / _fill / _transpose) and then runs an iteration as ONE
C call of two launches, `fmhr_stage1_step`: value and gradients of `pipeline.stage1_loss`, Adam optionally applied in the kernel.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib, dr, utils
from ._lib import call, ptr, stream_ptr


class HamStage1:
    def __init__(self, vertices, faces, w2cs, projs, imgs, masks, resolution, sfs_weight, chunk=8, want_pred_masks=False):
        """vertices [V,3], faces [F,3], w2cs / projs [num,4,4], imgs [num,H,W,3], masks [num,H,W] (all on the GPU).
        want_pred_masks: also keep the antialiased coverage of every view (`pred_mask` of mesh_sfs_optim.py:145-146, the start
        silhouettes stage 2 uses as `valid_masks`, :158) in self.pred_masks [num,H,W]."""
        _lib.require_gpu("HamStage1", vertices, w2cs, projs, imgs, masks)
        self.lib = _lib.load()
        dev = vertices.device
        self.dev, self.sfs_weight = dev, float(sfs_weight)
        H, W = resolution
        num, V = w2cs.shape[0], vertices.shape[0]
        self.n, self.V, self.H, self.W = num, V, H, W
        faces = faces.to(device=dev, dtype=torch.int32).contiguous()
        Fn = faces.shape[0]
        imgs = imgs.to(torch.float32).contiguous()
        masks = masks.to(torch.float32).contiguous()
        st = stream_ptr(dev)
        # ---- the constant sub-graph (:207-218), `chunk` views at a time: rast, normals, antialias work items
        rast = torch.empty(num, H, W, 4, dtype=torch.float32, device=dev)
        normals = torch.empty(num, H, W, 3, dtype=torch.float32, device=dev)
        items, alphas = [], []
        self.pred_masks = torch.empty(num, H, W, dtype=torch.float32, device=dev) if want_pred_masks else None
        glctx = dr.RasterizeGLContext()
        with torch.no_grad():
            for k in range(0, num, chunk):
                w2c, proj = w2cs[k:k + chunk], projs[k:k + chunk]
                c = w2c.shape[0]
                vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], dim=1).unsqueeze(0).expand(c, -1, -1)
                pos = torch.einsum('ijk,ikl->ijl', torch.einsum('ijk,ikl->ijl', vertsw, w2c), proj).contiguous()
                vn = utils.get_normals(vertsw[:, :, :3].contiguous(), faces.long())
                r, _ = dr.rasterize(glctx, pos, faces, resolution=resolution)
                if want_pred_masks:                                  # :143-146: normals and a ones channel through one interpolate
                    f4, _ = dr.interpolate(torch.cat([vn, torch.ones_like(vertsw[:, :, :1])], 2).contiguous(), r, faces)
                    feat = f4[..., :3]
                    self.pred_masks[k:k + c] = dr.antialias(f4[..., 3:4].contiguous(), r, pos, faces).squeeze(-1)
                else:
                    feat, _ = dr.interpolate(vn.contiguous(), r, faces)
                aa, it, al, overflow = dr.antialias_debug_items(feat.contiguous(), r, pos, faces)
                if overflow:
                    raise RuntimeError("HamStage1: antialias work-item list overflow")
                rast[k:k + c] = r
                normals[k:k + c] = F.normalize(aa, p=2, dim=3)
                it = it.clone()
                it[:, 2] += k                                        # view index within the chunk -> global
                items.append(it); alphas.append(al.clone())
        items = torch.cat(items).contiguous() if items else torch.zeros(0, 4, dtype=torch.int32, device=dev)
        alphas = torch.cat(alphas).contiguous() if alphas else torch.zeros(0, dtype=torch.float32, device=dev)
        n_items = int(items.shape[0])
        self.n_items = n_items
        # ---- the plan
        N = num * H * W
        i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        rowid, self.view_row0, self.view_valid = i32(N), i32(num + 1), i32(num)
        flags_len = N + num * 255
        row_flags = i32(flags_len)
        nb = self.lib.fmhr_stage1_rows_workspace_bytes(num, H, W)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        call("fmhr_stage1_rows", ptr(rast), ptr(masks), num, H, W, ptr(rowid), ptr(self.view_row0), ptr(self.view_valid), ptr(row_flags),
             flags_len, ptr(ws), nb, st)
        rows = int(self.view_row0[num])                               # host read (setup only)
        self.rows = rows
        self.row_ptr = i32(rows + 1)
        nb = self.lib.fmhr_stage1_row_ptr_workspace_bytes(rows)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        call("fmhr_stage1_row_ptr", ptr(rast), ptr(rowid), ptr(items), ptr(alphas), n_items, num, H, W, rows, ptr(row_flags),
             ptr(self.row_ptr), ptr(ws), nb, st)
        E = int(self.row_ptr[rows])
        self.entries = E
        self.ent_col, self.ent_w = i32(max(E, 1)), f32(max(E, 1))
        self.row_n, self.row_img = f32(max(rows, 1), 3), f32(max(rows, 1), 3)
        call("fmhr_stage1_fill", ptr(rast), ptr(faces), ptr(rowid), ptr(items), ptr(alphas), n_items, ptr(normals), ptr(imgs), num, V, Fn, H, W,
             rows, ptr(row_flags), ptr(self.row_ptr), ptr(self.ent_col), ptr(self.ent_w), ptr(self.row_n), ptr(self.row_img), st)
        self.t_ptr, self.t_row, self.t_w = i32(num * V + 1), i32(max(E, 1)), f32(max(E, 1))
        nb = self.lib.fmhr_stage1_transpose_workspace_bytes(num, V)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        call("fmhr_stage1_transpose", ptr(self.row_ptr), ptr(self.ent_col), ptr(self.ent_w), ptr(self.view_row0), num, V, rows,
             ptr(self.t_ptr), ptr(self.t_row), ptr(self.t_w), ptr(ws), nb, st)
        torch.cuda.synchronize(dev)                                   # the scratch tensors above go out of scope
        self.plan = _lib.Stage1Plan(num, V, rows, 0, self.view_row0.data_ptr(), self.view_valid.data_ptr(), self.row_ptr.data_ptr(),
                                    self.ent_col.data_ptr(), self.ent_w.data_ptr(), self.row_n.data_ptr(), self.row_img.data_ptr(),
                                    self.t_ptr.data_ptr(), self.t_row.data_ptr(), self.t_w.data_ptr())
        self.ws_bytes = self.lib.fmhr_stage1_step_workspace_bytes(ctypes.byref(self.plan))
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)
        self.galbedo, self.gsh, self.loss_view = f32(V, 3), f32(num, 9), f32(num)
        self.view_sel = torch.ones(num, dtype=torch.int32, device=dev)
        self.rast, self.normals, self.masks, self.imgs = rast, normals, masks, imgs      # the constant sub-graph (also read by init_sh)

    def step(self, albedo, sh_coeffs, idx=None, optimizer=None, params=None, sel=None):
        """One iteration on the views `idx` (None: all) — or on the views whose entry of `sel` (int32 [num] GPU tensor, the C ABI's own
        form: no index kernels in front of the step) is non-zero.  albedo [1,V,3] or [V,3], sh_coeffs [num,9].  Returns (loss_view [num],
        galbedo [V,3], gsh [num,9]) — device tensors, overwritten by the next call; the loss of mesh_sfs_optim.py:233 is
        loss_view.sum() (the views' shares; no extra launch unless somebody asks).  With `optimizer` (fmhr_amd.optim.Adam) and
        params = (albedo, sh_coeffs) the Adam update of both is applied by the step's second kernel (no optimizer.step())."""
        _lib.require_gpu("HamStage1.step", albedo, sh_coeffs)
        if not (albedo.is_contiguous() and sh_coeffs.is_contiguous() and albedo.dtype == torch.float32 and sh_coeffs.dtype == torch.float32):
            raise RuntimeError("HamStage1.step: albedo / sh_coeffs must be contiguous float32")
        if albedo.numel() != self.V * 3:
            raise RuntimeError(f"HamStage1.step: albedo must be [1,{self.V},3] or [{self.V},3]")
        a = albedo.detach().view(self.V, 3)                          # a view, never a copy: the kernel may update it in place
        s = sh_coeffs.detach()
        if s.shape != (self.n, 9):
            raise RuntimeError(f"HamStage1.step: sh_coeffs must be [{self.n}, 9]")
        if sel is not None:
            if idx is not None:
                raise RuntimeError("HamStage1.step: idx and sel are two forms of one selection: pass one")
            if sel.dtype != torch.int32 or not sel.is_cuda or tuple(sel.shape) != (self.n,) or not sel.is_contiguous():
                raise RuntimeError(f"HamStage1.step: sel must be a contiguous int32 GPU tensor of shape ({self.n},)")
        elif idx is not None:
            self.view_sel.zero_()
            self.view_sel[idx.long()] = 1
            sel = self.view_sel
        adam = (None, None)
        if optimizer is not None:
            if params is None or len(params) != 2:
                raise RuntimeError("HamStage1.step: optimizer= needs params=(albedo, sh_coeffs), the tensors the optimiser owns")
            if params[0].data_ptr() != a.data_ptr() or params[1].data_ptr() != s.data_ptr():
                raise RuntimeError("HamStage1.step: `params` must be the tensors behind the `albedo` / `sh_coeffs` arguments")
            adam = (optimizer.fused_param(params[0]), optimizer.fused_param(params[1]))
        try:
            call("fmhr_stage1_step", ctypes.byref(self.plan), ptr(a), ptr(s), ptr(sel), self.sfs_weight, ptr(self.galbedo), ptr(self.gsh),
                 ptr(self.loss_view), ctypes.byref(adam[0]) if adam[0] is not None else None,
                 ctypes.byref(adam[1]) if adam[1] is not None else None, ptr(self.ws), self.ws_bytes, stream_ptr(self.dev))
        except RuntimeError:
            if optimizer is not None:                                # a rejected call applied no update: take the step count back
                for p_ in params:
                    optimizer.state[p_]['step'] -= 1
            raise
        return self.loss_view, self.galbedo, self.gsh


def init_sh(hs1, grayimgs, nblk=64, group=None, view0=0, n_views_global=None, degree=3):
    """(view-sharded: `hs1` holds this rank's views [view0, view0 + hs1.n) of a capture of `n_views_global`, `grayimgs` their gray
    images, `group` the torch.distributed group of the ranks: the per-view solves stay local, the GLOBAL least squares of
    mesh_sfs_optim.py:165-166 and the mean albedo of :172-174 are solved from the all-reduced float64 sums — 55 + 4 numbers —, and the
    start silhouettes, which stage 2 reads for arbitrary views of every batch, are gathered to [n_views_global, H, W] on every rank.)
    `degree` 1 / 2 (conf key `degree`, models/utils.py:188-226): the design matrix of degree d is the first d^2 columns of the order-3 one
    (fmhr_amd.utils.get_matrix), so the degree-d least squares IS the leading d^2 x d^2 block of the accumulated normal equations; the
    returned rows have d^2 entries (`sh_coeffs`, `sh_global`), zero-padded to 9 where a kernel wants the order-3 form.
    The SH least-squares initialisation of mesh_sfs_optim.py:124-174 on the constant sub-graph a HamStage1 already holds (the same
    rasterization / antialiased normals / validity): per-view and global coefficients, the mean albedo.  The reference solves
    [Pv, 9] float64 least-squares problems with np.linalg.lstsq on the host, one view at a time; here the GPU accumulates the normal
    equations (fmhr_sh_normal_equations: 55 float64 sums per view) and only the 9 x 9 solves run on the host.
    Returns dict(sh_coeffs [num,9], sh_global [9], albedo_mean [3], valid_masks = hs1.pred_masks)."""
    import numpy as np
    dev, n, H, W = hs1.dev, hs1.n, hs1.H, hs1.W
    gray = grayimgs.to(torch.float32).contiguous()
    st = stream_ptr(dev)
    part = torch.empty(n, nblk, 55, dtype=torch.float64, device=dev)
    call("fmhr_sh_normal_equations", ptr(hs1.normals), ptr(hs1.rast), ptr(hs1.masks), ptr(gray), n, H, W, nblk, ptr(part), st)
    acc = part.sum(1).cpu().numpy()                                   # [n, 55]
    iu = np.triu_indices(9)
    if degree not in (1, 2, 3):
        raise RuntimeError("init_sh: degree must be 1, 2 or 3")
    d2 = degree * degree

    def solve(a, view=None):
        G = np.zeros((9, 9))
        G[iu] = a[:45]
        G = (G + np.triu(G, 1).T)[:d2, :d2]
        # The normal equations square the condition number: they reproduce np.linalg.lstsq on the [Pv, 9] design matrix (the
        # reference, mesh_sfs_optim.py:153-154,165-166) while cond(G) stays far from 1 / eps.  A view whose normals barely span the
        # basis (a flat patch, a handful of valid pixels) is solved the reference's way instead: its valid pixels are gathered and
        # handed to lstsq on the design matrix, whose rcond cut-off then acts on the design matrix's own singular values.
        sv = np.linalg.svd(G, compute_uv=False)
        if view is not None and (sv[-1] <= 1e-10 * sv[0]):
            valid = (hs1.masks[view] > 0) & (hs1.rast[view, :, :, 3] > 0)
            vn = hs1.normals[view][valid].cpu().numpy()
            if vn.shape[0] > 0:
                from .utils import get_matrix
                return np.linalg.lstsq(get_matrix(vn, degree), gray[view][valid].cpu().numpy(), rcond=None)[0]
        return np.linalg.lstsq(G, a[45:45 + d2], rcond=None)[0]
    sharded = group is not None and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1
    sh_views = np.stack([solve(acc[b], b) for b in range(n)]).astype(np.float32)
    acc_all = acc.sum(0)
    if sharded:
        t = torch.from_numpy(acc_all).to(dev)
        torch.distributed.all_reduce(t, group=group)
        acc_all = t.cpu().numpy()
    sh_global = solve(acc_all).astype(np.float32)
    shg = torch.from_numpy(sh_global).to(dev)
    shg9 = F.pad(shg, (0, 9 - d2)).contiguous()                       # the kernels evaluate the order-3 polynomial: higher bands zero
    part2 = torch.empty(n, nblk, 4, dtype=torch.float64, device=dev)
    call("fmhr_sh_albedo_sums", ptr(hs1.normals), ptr(hs1.rast), ptr(hs1.masks), ptr(hs1.imgs), ptr(shg9), n, H, W, nblk, ptr(part2), st)
    tot = part2.sum((0, 1))
    if sharded:
        torch.distributed.all_reduce(tot, group=group)
    tot = tot.cpu().numpy()
    valid_masks = hs1.pred_masks
    if sharded and valid_masks is not None:
        full = torch.zeros(int(n_views_global), H, W, dtype=torch.float32, device=dev)
        full[view0: view0 + n] = valid_masks
        torch.distributed.all_reduce(full, group=group)
        valid_masks = full
    albedo_mean = torch.from_numpy((tot[:3] / max(tot[3], 1.0)).astype(np.float32)).to(dev)
    return dict(sh_coeffs=torch.from_numpy(sh_views).to(dev), sh_global=shg, albedo_mean=albedo_mean, valid_masks=valid_masks)
