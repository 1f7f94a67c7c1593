"""fmhr_amd.dr — the nvdiffrast surface the reference uses (`import nvdiffrast.torch as dr`,
/root/reference/mesh_sfs_optim.py:14), backed by the gfx950 HIP kernels through the C ABI.

    glctx = dr.RasterizeGLContext()                                   # mesh_sfs_optim.py:120
    rast, _ = dr.rasterize(glctx, pos, tri, resolution=(H, W))        # :142, :212, :267
    out, _  = dr.interpolate(attr, rast, tri)                         # :143, :214, :269
    img     = dr.antialias(color, rast, pos, tri)                     # :146-147, :217-219, :274, :287

Same argument meaning, tuple arities and error behaviour (RuntimeError on bad shape/dtype/device); differentiable w.r.t.
pos / attr / rast[...,0:2] / color exactly as the callers rely on.  Rasterization rules: DESIGN.md §3.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


class RasterizeGLContext:
    """Opaque context (nvdiffrast's owns an OpenGL context; this one owns per-topology tables and scratch)."""

    def __init__(self, output_db=True, mode='automatic', device=None):
        self._topo_key = None
        self._tri_ref = None     # keeps the keyed tensor alive so its address cannot be reused by another tensor
        self._opp = None
        self._scratch = {}       # reusable scratch by (name, device, stream): the rasterizer's key / queue workspace

    def scratch(self, name, nbytes, device):
        """a uint8 scratch buffer of at least `nbytes`, owned by the context and reused by every call that asks for `name` ON THE SAME
        STREAM (round 5: a 48-view 512 x 512 rasterize allocated 138 MB of depth keys + queue per call; what a call leaves in it is dead
        when it returns).  Keyed by the current stream as well: two calls on different streams that share a context — the
        default context is global — get different buffers instead of racing on one; stream order protects a buffer within its stream."""
        key = (name, device, torch.cuda.current_stream(device).cuda_stream)
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self._scratch[key] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        return buf

    def opp_table(self, tri, V):
        """edge -> opposite-vertex table for antialias, rebuilt only when the triangle tensor changes."""
        key = (tri.data_ptr(), tri._version, tri.shape[0], int(V), tri.device)
        if key != self._topo_key:
            F = tri.shape[0]
            opp = torch.full((F, 3), -1, dtype=torch.int32, device=tri.device)
            nb = _lib.load().fmhr_topology_workspace_bytes(F)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=tri.device)
            call("fmhr_topology_build", ptr(tri), F, int(V), ptr(opp), ptr(ws), nb, stream_ptr(tri.device))
            self._topo_key, self._tri_ref, self._opp = key, tri, opp
        return self._opp


_default_ctx = None


def _ctx():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = RasterizeGLContext()
    return _default_ctx


def _check_tri(name, tri, device):
    if not isinstance(tri, torch.Tensor) or tri.dtype != torch.int32 or tri.dim() != 2 or tri.shape[1] != 3:
        raise RuntimeError(f"{name}: tri must be an int32 tensor of shape [F,3]")
    if tri.device != device:
        raise RuntimeError(f"{name}: tri must be on {device}")
    return tri.contiguous()


class _RasterizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W, glctx):
        n, V, _ = pos.shape
        F = tri.shape[0]
        rast = torch.empty(n, H, W, 4, dtype=torch.float32, device=pos.device)
        nb = _lib.load().fmhr_rasterize_workspace_bytes(n, V, F, H, W)
        ws = (glctx if isinstance(glctx, RasterizeGLContext) else _ctx()).scratch("rasterize", nb, pos.device)
        call("fmhr_rasterize_fwd", ptr(pos), ptr(tri), n, V, F, H, W, ptr(rast), ptr(ws), nb, stream_ptr(pos.device))
        ctx.save_for_backward(pos, tri, rast)
        return rast

    @staticmethod
    def backward(ctx, dy):
        pos, tri, rast = ctx.saved_tensors
        n, V, _ = pos.shape
        _, H, W, _ = rast.shape
        dy = dy.contiguous()
        gpos = torch.zeros_like(pos)
        call("fmhr_rasterize_bwd", ptr(pos), ptr(tri), ptr(rast), ptr(dy), n, V, tri.shape[0], H, W, ptr(gpos),
             stream_ptr(pos.device))
        return gpos, None, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """-> (rast [n,H,W,4] = (u, v, z/w, triangle_id+1), rast_db zeros).  `ranges` (range mode) is not used by FMHR."""
    if ranges is not None:
        raise RuntimeError("rasterize: range mode is not supported (FMHR only uses instanced mode)")
    if not isinstance(pos, torch.Tensor) or pos.dim() != 3 or pos.shape[2] != 4:
        raise RuntimeError("rasterize: pos must be a float32 tensor of shape [n,V,4] (instanced mode)")
    _lib.require_gpu("rasterize", pos)
    pos = _lib.f32c("rasterize", pos)
    tri = _check_tri("rasterize", tri, pos.device)
    if len(resolution) != 2 or int(resolution[0]) <= 0 or int(resolution[1]) <= 0:
        raise RuntimeError("rasterize: resolution must be (height, width) > 0")
    rast = _RasterizeFn.apply(pos, tri, int(resolution[0]), int(resolution[1]), glctx)
    # rast_db (the barycentrics' screen-space derivatives) is discarded at every call site of the reference (`rast_out, _ =`): a
    # zero tensor of the right shape WITHOUT its 16 B per pixel — one element expanded (a 48-view 512 x 512 call used to allocate
    # and zero-fill 200 MB here)
    return rast, torch.zeros((), dtype=rast.dtype, device=rast.device).expand(rast.shape)


class _InterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri):
        n, H, W, _ = rast.shape
        an, V, C = attr.shape
        out = torch.empty(n, H, W, C, dtype=torch.float32, device=attr.device)
        call("fmhr_interpolate_fwd", ptr(attr), ptr(rast), ptr(tri), n, an, V, tri.shape[0], H, W, C, ptr(out),
             stream_ptr(attr.device))
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
        call("fmhr_interpolate_bwd", ptr(attr), ptr(rast), ptr(tri), ptr(dy), n, an, V, tri.shape[0], H, W, C, ptr(gattr),
             ptr(grast), stream_ptr(attr.device))
        return gattr, grast, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """-> (out [n,H,W,C], out_da).  attr is [n,V,C] or [1,V,C]; pixel-differential outputs are not used by FMHR."""
    if rast_db is not None or diff_attrs is not None:
        raise RuntimeError("interpolate: rast_db / diff_attrs are not supported (never passed by FMHR)")
    if not isinstance(attr, torch.Tensor) or attr.dim() != 3:
        raise RuntimeError("interpolate: attr must be [n,V,C] or [1,V,C]")
    if rast.dim() != 4 or rast.shape[3] != 4:
        raise RuntimeError("interpolate: rast must be [n,H,W,4]")
    _lib.require_gpu("interpolate", attr, rast)
    if attr.shape[0] not in (1, rast.shape[0]):
        raise RuntimeError("interpolate: attr batch must be 1 or match rast")
    attr, rast = _lib.f32c("interpolate", attr), _lib.f32c("interpolate", rast)
    tri = _check_tri("interpolate", tri, attr.device)
    return _InterpolateFn.apply(attr, rast, tri), None


class _AntialiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, opp):
        n, H, W, C = color.shape
        V = pos.shape[1]
        dev = color.device
        out = torch.empty_like(color)
        # the work items are the pixel pairs that BLEND (one per silhouette crossing): a list of n*H*W / 8 entries (20 B each) holds any
        # silhouette a mesh can draw; the kernel flags an overflow instead of writing past it, and the call is repeated once with the
        # full bound (every horizontal + vertical pair) in that case.  (Round 4 allocated n*H*W entries — 252 MB at 48 x 512 x 512 — for
        # ~50 k items on EVERY call; the backward needs the list, so it cannot live in the context: two antialias calls per iteration
        # are alive at once)
        cap = max(n * H * W // 8, 4096)
        while True:
            items = torch.empty(cap, 4, dtype=torch.int32, device=dev)
            alphas = torch.empty(cap, dtype=torch.float32, device=dev)
            counter = torch.zeros(4, dtype=torch.int32, device=dev)
            call("fmhr_antialias_fwd", ptr(color), ptr(rast), ptr(pos), ptr(tri), ptr(opp), n, V, tri.shape[0], H, W, C, ptr(out),
                 ptr(items), ptr(alphas), ptr(counter), cap, stream_ptr(dev))
            if cap >= 2 * n * H * W or int(counter[1]) == 0:        # (the flag is read back: this op-by-op path synchronises per call anyway — its callers' torch.where does)
                break
            cap = 2 * n * H * W
        ctx.save_for_backward(color, rast, pos, tri, items, alphas, counter)
        ctx.cap = cap
        return out

    @staticmethod
    def backward(ctx, dy):
        color, rast, pos, tri, items, alphas, counter = ctx.saved_tensors
        n, H, W, C = color.shape
        V = pos.shape[1]
        dy = dy.contiguous()
        gcolor = torch.empty_like(color)
        gpos = torch.zeros_like(pos)
        call("fmhr_antialias_bwd", ptr(color), ptr(rast), ptr(pos), ptr(tri), ptr(dy), ptr(items), ptr(alphas), ptr(counter),
             ctx.cap, n, V, tri.shape[0], H, W, C, ptr(gcolor), ptr(gpos), stream_ptr(color.device))
        return gcolor, None, gpos, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """-> antialiased color [n,H,W,C].  The edge table is cached per triangle tensor (`topology_hash` is accepted and
    ignored: nvdiffrast rebuilds its hash on every call when it is None, as FMHR does)."""
    if color.dim() != 4 or rast.dim() != 4 or rast.shape[3] != 4 or color.shape[:3] != rast.shape[:3]:
        raise RuntimeError("antialias: color must be [n,H,W,C] and rast [n,H,W,4] with matching n,H,W")
    if pos.dim() != 3 or pos.shape[2] != 4 or pos.shape[0] != color.shape[0]:
        raise RuntimeError("antialias: pos must be [n,V,4]")
    if pos_gradient_boost != 1.0:
        raise RuntimeError("antialias: pos_gradient_boost != 1 is not supported")
    _lib.require_gpu("antialias", color, rast, pos)
    color, rast, pos = _lib.f32c("antialias", color), _lib.f32c("antialias", rast), _lib.f32c("antialias", pos)
    tri = _check_tri("antialias", tri, color.device)
    opp = _ctx().opp_table(tri, pos.shape[1])
    return _AntialiasFn.apply(color, rast, pos, tri, opp)


def antialias_debug_items(color, rast, pos, tri):
    """Test hook: run the forward and return (out, items [k,4], alphas [k], overflow flag)."""
    tri = _check_tri("antialias", tri, color.device)
    opp = _ctx().opp_table(tri, pos.shape[1])
    n, H, W, C = color.shape
    dev = color.device
    out = torch.empty_like(color)
    cap = max(n * H * W, 1)
    items = torch.empty(cap, 4, dtype=torch.int32, device=dev)
    alphas = torch.empty(cap, dtype=torch.float32, device=dev)
    counter = torch.zeros(4, dtype=torch.int32, device=dev)
    call("fmhr_antialias_fwd", ptr(color.contiguous()), ptr(rast.contiguous()), ptr(pos.contiguous()), ptr(tri), ptr(opp), n,
         pos.shape[1], tri.shape[0], H, W, C, ptr(out), ptr(items), ptr(alphas), ptr(counter), cap, stream_ptr(dev))
    k = int(counter[0])
    return out, items[:k], alphas[:k], int(counter[1])
