"""oracle/ham_oracle.py — TEST INFRASTRUCTURE (see oracle/README.md).  NOT imported by fmhr_amd.

CPU restatement of the HAM stage-2 hot path of the reference:

* the three nvdiffrast primitives (C, `raster_oracle.c`) wrapped as CPU ``torch.autograd.Function``s behind the same
  ``RasterizeGLContext / rasterize / interpolate / antialias`` surface the reference uses
  (/root/reference/mesh_sfs_optim.py:120,267,269,274,287)  — **parity unpinned** against real nvdiffrast;
* the torch helpers of /root/reference/models/utils.py (`get_normals` :508-548, `get_matrix` :188-206,
  `get_radiance` :208-226, `get_edges` :551-571, `compute_laplacian` :661-693, `laplacian_smoothing` :696-722) and
  `NCC` (/root/reference/models/ncc_utils.py:4-35) — pinned by tests/golden/*.npz generated from the imported reference;
* the stage-2 loop body /root/reference/mesh_sfs_optim.py:253-306 (`ham_stage2_losses`).

Works in float32 (decision-exact twin of the HIP kernels) and float64 (finite-difference target).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# "native": fp64 tensors take their discrete decisions in fp64.  "f32": fp64 tensors take the fp32 twin's decisions (the ones
# the HIP kernels take) and only the VALUES are fp64 — the reference for value tolerances (raster_oracle.c "mix" variants).
DECISIONS = "native"


class f32_decisions:
    """context manager: with ham_oracle.f32_decisions(): ... (fp64 values on fp32 decisions)"""

    def __enter__(self):
        global DECISIONS
        self.prev, DECISIONS = DECISIONS, "f32"

    def __exit__(self, *a):
        global DECISIONS
        DECISIONS = self.prev



def build():
    """Compile oracle/libraster_oracle.so with gcc (test infrastructure build; not the product build)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libraster_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < max(
                os.path.getmtime(os.path.join(_HERE, f)) for f in ("raster_oracle.c", "raster_oracle_impl.h")):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_topology_build.restype = ctypes.c_int
        for s in ("f32", "f64", "mix"):
            getattr(_LIB, "orc_antialias_fwd_" + s).restype = ctypes.c_int
        set_threads()
    return _LIB


def set_threads(n=None):
    """threads of the C ops' loops over views (OpenMP; results do not depend on the count).  Default: the CPUs this process may run
    on, at most 16 (a one-GPU job of the pool has a 16-CPU share whatever os.cpu_count() says); FMHR_CPU_THREADS overrides."""
    if n is None:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        n = int(os.environ.get("FMHR_CPU_THREADS", min(avail, 16)))
    lib().orc_set_num_threads(int(n))
    return int(lib().orc_get_max_threads())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _sfx(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise RuntimeError("oracle: float32/float64 only")


def _chk_tri(tri):
    if tri.dtype != torch.int32 or tri.dim() != 2 or tri.shape[1] != 3:
        raise RuntimeError("tri must be int32 [F,3]")
    return tri.contiguous()


def topology_build(tri, V):
    """opp[f,k]: the vertex opposite edge k (= edge (v[k+1],v[k+2])) in the neighbouring face, -1 on a boundary."""
    tri = _chk_tri(tri)
    opp = torch.full_like(tri, -1)
    rc = lib().orc_topology_build(_p(tri), tri.shape[0], int(V), _p(opp))
    if rc != 0:
        raise RuntimeError("orc_topology_build failed")
    return opp


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W):
        pos = pos.contiguous()
        n, V, _ = pos.shape
        rast = torch.empty(n, H, W, 4, dtype=pos.dtype)
        sfx = "mix" if (pos.dtype == torch.float64 and DECISIONS == "f32") else _sfx(pos)
        getattr(lib(), "orc_rasterize_fwd_" + sfx)(_p(pos), _p(tri), n, V, tri.shape[0], H, W, _p(rast))
        ctx.save_for_backward(pos, tri, rast)
        return rast

    @staticmethod
    def backward(ctx, dy):
        pos, tri, rast = ctx.saved_tensors
        n, V, _ = pos.shape
        _, H, W, _ = rast.shape
        dy = dy.contiguous()
        gpos = torch.zeros_like(pos)
        getattr(lib(), "orc_rasterize_bwd_" + _sfx(pos))(_p(pos), _p(tri), _p(rast), _p(dy), n, V, tri.shape[0], H, W, _p(gpos))
        return gpos, None, None, None


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        attr = attr.contiguous()
        rast = rast.contiguous()
        n, H, W, _ = rast.shape
        an, V, C = attr.shape
        out = torch.empty(n, H, W, C, dtype=attr.dtype)
        getattr(lib(), "orc_interpolate_fwd_" + _sfx(attr))(_p(attr), _p(rast), _p(tri), n, an, V, tri.shape[0], H, W, C, _p(out))
        ctx.save_for_backward(attr, rast, tri)
        return out

    @staticmethod
    def backward(ctx, dy):
        attr, rast, tri = ctx.saved_tensors
        n, H, W, _ = rast.shape
        an, V, C = attr.shape
        dy = dy.contiguous()
        gattr = torch.zeros_like(attr)
        grast = torch.empty_like(rast)
        getattr(lib(), "orc_interpolate_bwd_" + _sfx(attr))(_p(attr), _p(rast), _p(tri), _p(dy), n, an, V, tri.shape[0], H, W, C,
                                                            _p(gattr), _p(grast))
        return gattr, grast, None


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp):
        color = color.contiguous()
        rast = rast.contiguous()
        pos = pos.contiguous()
        n, H, W, C = color.shape
        V = pos.shape[1]
        out = torch.empty_like(color)
        max_items = 2 * n * H * W
        items = torch.empty(max_items, 4, dtype=torch.int32)
        alphas = torch.empty(max_items, dtype=color.dtype)
        sfx = "mix" if (color.dtype == torch.float64 and DECISIONS == "f32") else _sfx(color)
        cnt = getattr(lib(), "orc_antialias_fwd_" + sfx)(_p(color), _p(rast), _p(pos), _p(tri), _p(opp), n, V, tri.shape[0],
                                                                 H, W, C, _p(out), _p(items), _p(alphas), max_items)
        items = items[:cnt].clone()
        alphas = alphas[:cnt].clone()
        ctx.save_for_backward(color, rast, pos, tri, items, alphas)
        ctx.mark_non_differentiable(items, alphas)
        return out, items, alphas

    @staticmethod
    def backward(ctx, dy, _gi, _ga):
        color, rast, pos, tri, items, alphas = ctx.saved_tensors
        n, H, W, C = color.shape
        V = pos.shape[1]
        dy = dy.contiguous()
        gcolor = torch.empty_like(color)
        gpos = torch.zeros_like(pos)
        getattr(lib(), "orc_antialias_bwd_" + _sfx(color))(_p(color), _p(rast), _p(pos), _p(tri), _p(dy), _p(items), _p(alphas),
                                                           items.shape[0], n, V, tri.shape[0], H, W, C, _p(gcolor), _p(gpos))
        return gcolor, None, gpos, None, None


# ------------------------------------------------------------------ nvdiffrast-shaped surface (the `dr` module)
class RasterizeGLContext:
    """Stateless handle; caches the edge->opposite-vertex table per topology."""

    def __init__(self):
        self._topo = {}

    def opp(self, tri, V):
        key = (tri.data_ptr(), tri._version, tri.shape[0], V)
        if key not in self._topo:
            self._topo = {key: (topology_build(tri, V), tri)}     # keep `tri` alive: its address is part of the key
        return self._topo[key][0]


_DEFAULT_CTX = RasterizeGLContext()


def rasterize(glctx, pos, tri, resolution):
    H, W = int(resolution[0]), int(resolution[1])
    if pos.dim() != 3 or pos.shape[2] != 4:
        raise RuntimeError("pos must be [n,V,4]")
    rast = _Rasterize.apply(pos, _chk_tri(tri), H, W)
    return rast, torch.zeros_like(rast)


def interpolate(attr, rast, tri):
    if attr.dim() != 3:
        raise RuntimeError("attr must be [n,V,C] or [1,V,C]")
    return _Interpolate.apply(attr, rast, _chk_tri(tri)), None


def antialias(color, rast, pos, tri, return_items=False):
    tri = _chk_tri(tri)
    opp = _DEFAULT_CTX.opp(tri, pos.shape[1])
    out, items, alphas = _Antialias.apply(color, rast, pos, tri, opp)
    if return_items:
        return out, items, alphas
    return out


# ------------------------------------------------------------------ helpers of models/utils.py, restated
def get_normals(vertices, faces):
    """Area-weighted vertex normals, [b,n,3] -> [b,n,3] (models/utils.py:508-548): each face adds the cross product of the
    two edges leaving corner k to vertex k (accumulation order 1, 2, 0), then x / max(|x|, 1e-6)."""
    faces = faces.long()
    acc = torch.zeros_like(vertices)
    corner = vertices[:, faces]  # b f 3 3
    for k in (1, 2, 0):
        e1 = corner[:, :, (k + 1) % 3] - corner[:, :, k]
        e2 = corner[:, :, (k + 2) % 3] - corner[:, :, k]
        acc = acc.index_add(1, faces[:, k], torch.cross(e1, e2, dim=2))
    nrm = acc.norm(dim=2, keepdim=True).clamp_min(1e-6)
    return acc / nrm


def get_matrix(normal, degree=3):
    """SH design matrix, rows (1, y, z, x, xy, yz, 2z^2-x^2-y^2, zx, x^2-y^2) (models/utils.py:188-206)."""
    is_np = isinstance(normal, np.ndarray)
    n = torch.as_tensor(normal)
    x, y, z = n[:, 0], n[:, 1], n[:, 2]
    cols = [torch.ones_like(x)]
    if degree > 1:
        cols += [y, z, x]
    if degree > 2:
        cols += [x * y, y * z, 2 * z * z - x * x - y * y, z * x, x * x - y * y]
    m = torch.stack(cols, dim=1)
    return m.numpy() if is_np else m


def get_radiance(coeff, normal, degree=3):
    """r = sum_k coeff[..., k] * basis_k(normal) (models/utils.py:208-226); coeff is [9] or [n,9]."""
    m = get_matrix(normal, degree)
    return (coeff[..., : degree * degree] * m).sum(-1)


def get_edges(verts, faces):
    """Unique undirected edges sorted by hash V*min+max, and faces_to_edges with column order (e12, e20, e01)
    (models/utils.py:551-571)."""
    V = verts.shape[0]
    f = faces.long()
    pairs = torch.cat([f[:, [1, 2]], f[:, [2, 0]], f[:, [0, 1]]], dim=0)
    lo = pairs.min(dim=1).values
    hi = pairs.max(dim=1).values
    h, inv = torch.unique(V * lo + hi, return_inverse=True)
    edges = torch.stack([h // V, h % V], dim=1)
    return edges, inv.reshape(3, f.shape[0]).t()


def compute_laplacian(verts, faces):
    """Uniform Laplacian as a dense-free operator (models/utils.py:661-693): L_ij = 1/deg(i) on edges, L_ii = -1.
    Returns a sparse COO tensor [V,V]."""
    V = verts.shape[0]
    edges, _ = get_edges(verts, faces)
    e0, e1 = edges[:, 0], edges[:, 1]
    deg = torch.zeros(V, dtype=verts.dtype).index_add_(0, e0, torch.ones(e0.shape[0], dtype=verts.dtype))
    deg = deg.index_add_(0, e1, torch.ones(e1.shape[0], dtype=verts.dtype))
    inv = torch.where(deg > 0, 1.0 / deg.clamp_min(1), deg)
    rows = torch.cat([e0, e1, torch.arange(V)])
    cols = torch.cat([e1, e0, torch.arange(V)])
    vals = torch.cat([inv[e0], inv[e1], -torch.ones(V, dtype=verts.dtype)])
    return torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (V, V)).coalesce()


def laplacian_smoothing(verts, faces, method="uniform"):
    """(1/V) * sum_i |(L x)_i|_2 (models/utils.py:696-722, method="uniform" is the only one on the hot path)."""
    if method != "uniform":
        raise NotImplementedError("only the uniform Laplacian is on the HAM hot path")
    with torch.no_grad():
        L = compute_laplacian(verts, faces)
    lx = torch.sparse.mm(L, verts)
    return lx.norm(dim=1).sum() * (1.0 / verts.shape[0])


def NCC(ref, src, ref_valid_mask, src_valid_mask):
    """Masked normalised cross-correlation (models/ncc_utils.py:4-35); ref_valid_mask is accepted and unused there too."""
    m = src_valid_mask
    cnt = m.sum(dim=2, keepdim=True)
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
    r = ref.expand(src.shape[0], -1, -1)

    def stats(x):
        mean = (x * m).sum(dim=2, keepdim=True) / cnt
        var = (((x - mean) * m) ** 2).sum(dim=2, keepdim=True) / cnt
        var = var + (var == 0).to(var.dtype)
        return mean, var.sqrt()

    rm, rs = stats(r)
    sm, ss = stats(src)
    cov = ((r - rm) * (src - sm) * m).sum(dim=2, keepdim=True) / cnt
    return (cov / (rs * ss)).squeeze()


# ------------------------------------------------------------------ the stage-2 loop body, mesh_sfs_optim.py:253-306
def clip_positions(vertices, w2c, proj):
    """Row-vector clip transform of mesh_sfs_optim.py:262-264: pos = [v,1] @ w2c @ proj, per view."""
    n = w2c.shape[0]
    vertsw = torch.cat([vertices, torch.ones_like(vertices[:, 0:1])], dim=1).unsqueeze(0).expand(n, -1, -1)
    rot = torch.einsum('ijk,ikl->ijl', vertsw, w2c)
    return vertsw, torch.einsum('ijk,ikl->ijl', rot, proj)


def ham_stage2_losses(vertices_tmp, delta, albedo, sh_coeff, w2c, proj, img, mask, valid_mask, faces, resolution,
                      weights, edge_length_mean, degree=3, sfs_denominator=None, mse_denominator=None, dr=None,
                      pos_value=None, decisions=None):
    """One stage-2 forward of the reference (mesh_sfs_optim.py:253-306) on top of the oracle `dr` ops.

    weights: dict(sfs, lap, albedo, mask, edge, delta).  Returns dict with the six losses, `loss`, and the named
    intermediates.  `sfs_denominator` / `mse_denominator` override the batch-local divisors 3*Pv and n*P (used by the
    view-sharded path).  `pos_value` injects clip positions (values only).
    `decisions`: the `decisions` entry a previous fp32 run of the same inputs returned.  The loop body has two discrete decisions of
    its own besides the rasterizer's (which the C library's "mix" variants take from the fp32 twin under f32_decisions): the SIGN of
    every L1 residual (:289, d |d| = sign(d)) and the ACTIVE SET of the edge hinge (:302, clip(x, 0, 1) has a gradient only for
    0 < x < 1).  A residual or a hinge argument within fp32 rounding of its threshold is decided differently by exact arithmetic —
    one flipped edge of 295 k moves the gradient by 7e-4 of its maximum on the sub3 mesh — so "fp64 values on fp32 decisions"
    takes these from the fp32 run as well: sum(sign32 * d64) and where(active32, x64, const)."""
    import torch.nn.functional as F
    import sys
    dr = dr or sys.modules[__name__]
    glctx = dr.RasterizeGLContext()
    vertices = vertices_tmp + delta
    n = w2c.shape[0]
    vertsw, proj_verts = clip_positions(vertices, w2c, proj)
    if pos_value is not None:
        # evaluate at externally supplied clip positions (e.g. the GPU's, which differ in the last bit from torch's einsum)
        # while keeping the autograd graph: a + (b - a) == b exactly for nearby floats
        proj_verts = proj_verts + (pos_value.to(proj_verts.dtype) - proj_verts).detach()
    normals = get_normals(vertsw[:, :, :3], faces.long())
    rast_out, _ = dr.rasterize(glctx, proj_verts, faces, resolution=resolution)
    feat = torch.cat([normals, albedo.expand(n, -1, -1), torch.ones_like(vertsw[:, :, :1])], dim=2)
    feat, _ = dr.interpolate(feat, rast_out, faces)
    pred_normals = F.normalize(feat[:, :, :, :3].contiguous(), p=2, dim=3)
    rast_albedo = feat[:, :, :, 3:6].contiguous()
    pred_mask = dr.antialias(feat[:, :, :, 6:7].contiguous(), rast_out, proj_verts, faces).squeeze(-1)
    valid = (mask > 0) & (rast_out[:, :, :, 3] > 0)
    valid_idx = torch.where(valid)
    radiance = get_radiance(sh_coeff[valid_idx[0]], pred_normals[valid_idx], degree).unsqueeze(-1)
    pred_img = radiance * rast_albedo[valid_idx]
    tmp_img = torch.zeros_like(img)
    tmp_img[valid_idx] = pred_img
    tmp_img = dr.antialias(tmp_img, rast_out, proj_verts, faces)
    resid = tmp_img[valid_idx] - img[valid_idx]
    l1_sign = torch.sign(resid.detach())
    if decisions is not None:
        l1_sign = decisions["l1_sign"].to(resid.dtype)
        assert l1_sign.shape == resid.shape, "decisions come from a run with another valid set"
        diff = l1_sign * resid                                   # == |resid| wherever the two runs agree on the sign
    else:
        diff = resid.abs()
    den = diff.numel() if sfs_denominator is None else sfs_denominator
    sfs_loss = weights["sfs"] * diff.sum() / den
    lap_loss = weights["lap"] * laplacian_smoothing(vertices, faces.long(), method="uniform")
    albedo_loss = weights["albedo"] * laplacian_smoothing(albedo.squeeze(0), faces.long(), method="uniform")
    sq = (pred_mask - valid_mask) ** 2
    mask_loss = weights["mask"] * sq.sum() / (sq.numel() if mse_denominator is None else mse_denominator)
    fl = faces.long()
    a, b, c = vertices[fl[:, 0]], vertices[fl[:, 1]], vertices[fl[:, 2]]
    edge_length = torch.cat([((a - b) ** 2).sum(1), ((c - b) ** 2).sum(1), ((a - c) ** 2).sum(1)])
    hinge_x = edge_length - edge_length_mean
    hinge_active = ((hinge_x > 0) & (hinge_x < 1)).detach()
    if decisions is not None:
        hinge_active = decisions["hinge_active"]
        edge_loss = torch.where(hinge_active, hinge_x, torch.clip(hinge_x, 0, 1).detach()).mean() * weights["edge"]
    else:
        edge_loss = torch.clip(hinge_x, 0, 1).mean() * weights["edge"]
    delta_loss = (delta ** 2).sum(1).mean() * weights["delta"]
    loss = sfs_loss + lap_loss + albedo_loss + mask_loss + delta_loss + edge_loss
    return dict(loss=loss, sfs=sfs_loss, lap=lap_loss, albedo=albedo_loss, mask=mask_loss, edge=edge_loss, delta=delta_loss,
                pos=proj_verts, normals=normals, rast=rast_out, pred_normals=pred_normals, pred_mask=pred_mask,
                tmp_img=tmp_img, n_valid=int(valid.sum()), decisions=dict(l1_sign=l1_sign, hinge_active=hinge_active))


def edge_length_mean(vertices, faces):
    """mean squared length of the 3F directed face edges of the initial mesh (mesh_sfs_optim.py:184-188)."""
    fl = faces.long()
    a, b, c = vertices[fl[:, 0]], vertices[fl[:, 1]], vertices[fl[:, 2]]
    return torch.cat([((a - b) ** 2).sum(1), ((c - b) ** 2).sum(1), ((a - c) ** 2).sum(1)]).mean()
