"""fmhr_amd.utils — the helpers of /root/reference/models/utils.py that sit on the HAM hot path, with the same names and
argument meaning, backed by gfx950 kernels (C ABI).  GPU tensors only; there is no CPU path.

    get_normals(vertices[b,n,3], faces[f,3])            models/utils.py:508-548
    get_matrix(normal[n,3], degree=3)                   models/utils.py:188-206   (design matrix of the SH basis)
    get_radiance(coeff[9] or [n,9], normal[n,3], 3)     models/utils.py:208-226
    laplacian_smoothing(verts[V,C], faces, "uniform")   models/utils.py:696-722
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def _faces_i32(faces):
    if faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("faces must be [F,3]")
    if faces.dtype == torch.int32:
        return faces.contiguous()
    if faces.dtype == torch.int64:
        return faces.to(torch.int32).contiguous()
    raise RuntimeError("faces must be int32 or int64")


class _NormalsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, tri):
        b, V, _ = vertices.shape
        nsum = torch.empty_like(vertices)
        normals = torch.empty_like(vertices)
        call("fmhr_vertex_normals_fwd", ptr(vertices), ptr(tri), b, V, tri.shape[0], ptr(nsum), ptr(normals),
             stream_ptr(vertices.device))
        ctx.save_for_backward(vertices, tri, nsum)
        return normals

    @staticmethod
    def backward(ctx, g):
        vertices, tri, nsum = ctx.saved_tensors
        b, V, _ = vertices.shape
        g = g.contiguous()
        gtmp = torch.empty_like(vertices)
        gverts = torch.zeros_like(vertices)
        call("fmhr_vertex_normals_bwd", ptr(vertices), ptr(tri), ptr(nsum), ptr(g), b, V, tri.shape[0], ptr(gtmp), ptr(gverts),
             stream_ptr(vertices.device))
        return gverts, None


def get_normals(vertices, faces):
    """vertices b n 3, faces f 3 -> area-weighted unit vertex normals b n 3."""
    if vertices.dim() != 3 or vertices.shape[2] != 3:
        raise RuntimeError("get_normals: vertices must be [b,n,3]")
    _lib.require_gpu("get_normals", vertices, faces)
    if vertices.shape[0] > 1 and vertices.stride(0) == 0:
        # the reference hands over the SAME world-space vertices n times (`vertsw = ....unsqueeze(0).expand(n, -1, -1)`,
        # mesh_sfs_optim.py:262,265): the normals are view-independent, so they are formed once and the result is expanded — the
        # values of every view are the ones the batched call would give, autograd sums the n gradient slices back through the expand
        # (k_normals_faces 0.83 -> 0.02 ms, k_normals_bwd_faces 0.42 -> 0.01 ms per 48-view iteration of the unchanged loop body)
        one = _NormalsFn.apply(_lib.f32c("get_normals", vertices[0:1]), _faces_i32(faces))
        return one.expand(vertices.shape[0], -1, -1)
    return _NormalsFn.apply(_lib.f32c("get_normals", vertices), _faces_i32(faces))


def get_matrix(normal, degree=3):
    """SH design matrix [n, degree^2], columns (1, y, z, x, xy, yz, 2z^2-x^2-y^2, zx, x^2-y^2).  Plain array algebra: it
    is only used for the one-off least-squares initialisation, where the reference hands it a NumPy array of the valid pixels'
    normals (mesh_sfs_optim.py:153,165) — np.ndarray in -> float64 np.ndarray out, exactly as models/utils.py:188-190 (the
    products are formed in the input's precision and stored into a float64 matrix); torch.Tensor in -> tensor out."""
    if isinstance(normal, np.ndarray):
        matrix = np.zeros((normal.shape[0], degree ** 2))
        matrix[:, 0] = 1
        if degree > 1:
            matrix[:, 1], matrix[:, 2], matrix[:, 3] = normal[:, 1], normal[:, 2], normal[:, 0]
        if degree > 2:
            x, y, z = normal[:, 0], normal[:, 1], normal[:, 2]
            matrix[:, 4] = x * y
            matrix[:, 5] = y * z
            matrix[:, 6] = 2 * z * z - x * x - y * y
            matrix[:, 7] = z * x
            matrix[:, 8] = x * x - y * y
        return matrix
    if not isinstance(normal, torch.Tensor):
        raise RuntimeError("get_matrix: normal must be a numpy array or a torch tensor [n,3]")
    x, y, z = normal[:, 0], normal[:, 1], normal[:, 2]
    cols = [torch.ones_like(x)]
    if degree > 1:
        cols += [y, z, x]
    if degree > 2:
        cols += [x * y, y * z, 2 * z * z - x * x - y * y, z * x, x * x - y * y]
    return torch.stack(cols, dim=1)


class _RadianceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coeff, normal):
        P = normal.shape[0]
        per_pixel = 1 if coeff.dim() == 2 else 0
        out = torch.empty(P, dtype=torch.float32, device=normal.device)
        call("fmhr_sh_radiance_fwd", ptr(coeff), ptr(normal), P, per_pixel, ptr(out), stream_ptr(normal.device))
        ctx.save_for_backward(coeff, normal)
        return out

    @staticmethod
    def backward(ctx, dy):
        coeff, normal = ctx.saved_tensors
        P = normal.shape[0]
        per_pixel = 1 if coeff.dim() == 2 else 0
        dy = dy.contiguous()
        gcoeff = torch.zeros_like(coeff)
        gnormal = torch.empty_like(normal)
        call("fmhr_sh_radiance_bwd", ptr(coeff), ptr(normal), ptr(dy), P, per_pixel, ptr(gcoeff), ptr(gnormal),
             stream_ptr(normal.device))
        return gcoeff, gnormal


def get_radiance(coeff, normal, degree=3):
    """coeff degree^2 or n degree^2, normal n 3 -> radiance n (SH irradiance polynomial without normalisation constants;
    degree 3 on the HAM hot path, conf/demo_sfs.conf:29; degrees 1 and 2 are the same polynomial with the higher bands' coefficients
    zero: the coefficient rows are zero-padded to 9, autograd slices the gradient back)."""
    if degree not in (1, 2, 3):
        raise RuntimeError("get_radiance: degree must be 1, 2 or 3")
    if degree < 3:
        if coeff.shape[-1] < degree ** 2:
            raise RuntimeError("get_radiance: coeff must have at least degree^2 entries")
        coeff = torch.nn.functional.pad(coeff[..., :degree ** 2], (0, 9 - degree ** 2))
    if normal.dim() != 2 or normal.shape[1] != 3 or coeff.shape[-1] != 9 or coeff.dim() not in (1, 2):
        raise RuntimeError("get_radiance: coeff must be [9] or [n,9], normal [n,3]")
    if coeff.dim() == 2 and coeff.shape[0] != normal.shape[0]:
        raise RuntimeError("get_radiance: per-pixel coeff must have one row per normal")
    _lib.require_gpu("get_radiance", coeff, normal)
    return _RadianceFn.apply(_lib.f32c("get_radiance", coeff), _lib.f32c("get_radiance", normal))


class MeshAdjacency:
    """Unique-neighbour CSR of a triangle mesh, built on the GPU (replaces compute_laplacian's per-call rebuild)."""

    def __init__(self, tri, V):
        F = tri.shape[0]
        dev = tri.device
        self.tri_ref = tri
        self.V = int(V)
        self.row_ptr = torch.zeros(V + 1, dtype=torch.int32, device=dev)
        self.col_idx = torch.zeros(max(6 * F, 1), dtype=torch.int32, device=dev)
        nb = _lib.load().fmhr_adjacency_workspace_bytes(int(V), F)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        call("fmhr_adjacency_build", ptr(tri), F, int(V), ptr(self.row_ptr), ptr(self.col_idx), ptr(ws), nb, stream_ptr(dev))


_adj_cache = {"key": None, "adj": None}


def mesh_adjacency(faces, V):
    tri = _faces_i32(faces)
    key = (faces.data_ptr(), faces._version, faces.shape[0], int(V), faces.device)
    if _adj_cache["key"] != key:
        _adj_cache["key"], _adj_cache["adj"], _adj_cache["faces_ref"] = key, MeshAdjacency(tri, V), faces
    return _adj_cache["adj"]


def compute_laplacian(verts, faces):
    """The uniform Laplacian as a sparse COO tensor (L_ij = 1/deg(i), L_ii = -1), for callers that want the matrix."""
    V = verts.shape[0]
    adj = mesh_adjacency(faces, V)
    rp = adj.row_ptr.long()
    deg = (rp[1:] - rp[:-1])
    nnz = int(rp[-1])
    rows = torch.repeat_interleave(torch.arange(V, device=verts.device), deg)
    cols = adj.col_idx[:nnz].long()
    vals = (1.0 / deg.clamp_min(1).to(torch.float32))[rows]
    ar = torch.arange(V, device=verts.device)
    idx = torch.cat([torch.stack([rows, cols]), torch.stack([ar, ar])], 1)
    return torch.sparse_coo_tensor(idx, torch.cat([vals, -torch.ones(V, device=verts.device)]), (V, V)).coalesce()


class _LaplacianFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj):
        V, C = x.shape
        lx = torch.empty_like(x)
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        call("fmhr_laplacian_uniform_fwd", ptr(x), ptr(adj.row_ptr), ptr(adj.col_idx), V, C, 1.0, ptr(lx), ptr(loss),
             stream_ptr(x.device))
        ctx.save_for_backward(lx)
        ctx.adj = adj
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (lx,) = ctx.saved_tensors
        V, C = lx.shape
        rowg = torch.empty_like(lx)
        gx = torch.zeros_like(lx)
        call("fmhr_laplacian_uniform_bwd", ptr(lx), ptr(ctx.adj.row_ptr), ptr(ctx.adj.col_idx), V, C, 1.0, ptr(rowg), ptr(gx),
             stream_ptr(lx.device))
        return gx * g, None


def laplacian_smoothing(verts, faces, method="uniform"):
    """(1/V) sum_i |(L verts)_i|_2 with the uniform Laplacian (the only method the HAM loop uses)."""
    if method != "uniform":
        raise RuntimeError('laplacian_smoothing: only method="uniform" is on the HAM hot path')
    if verts.dim() != 2 or not (1 <= verts.shape[1] <= 4):
        raise RuntimeError("laplacian_smoothing: verts must be [V,C] with C <= 4")
    _lib.require_gpu("laplacian_smoothing", verts, faces)
    return _LaplacianFn.apply(_lib.f32c("laplacian_smoothing", verts), mesh_adjacency(faces, verts.shape[0]))
