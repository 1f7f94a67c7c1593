"""fmhr_amd.ncc_utils — `NCC` with the signature of /root/reference/models/ncc_utils.py:4-35, backed by a gfx950 kernel
(one wave64 per (view, point) patch).  GPU tensors only."""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


class _NccFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ref, src, mask):
        nv, Np, Npx = src.shape
        out = torch.empty(nv, Np, dtype=torch.float32, device=src.device)
        call("fmhr_ncc_fwd", ptr(ref), ptr(src), ptr(mask), nv, Np, Npx, ptr(out), stream_ptr(src.device))
        ctx.save_for_backward(ref, src, mask)
        return out

    @staticmethod
    def backward(ctx, dy):
        ref, src, mask = ctx.saved_tensors
        nv, Np, Npx = src.shape
        dy = dy.contiguous()
        gref = torch.zeros_like(ref) if ctx.needs_input_grad[0] else None
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[1] else None
        if gref is None and gsrc is None:
            return None, None, None
        call("fmhr_ncc_bwd", ptr(ref), ptr(src), ptr(mask), ptr(dy), nv, Np, Npx, ptr(gref), ptr(gsrc), stream_ptr(src.device))
        return gref, gsrc, None


def NCC(ref, src, ref_valid_mask, src_valid_mask):
    """ref 1 npoints npixels, src nviews npoints npixels, masks alike -> ncc nviews npoints (squeezed like the reference).
    `ref_valid_mask` is accepted and unused, exactly as in the reference."""
    if ref.dim() != 3 or src.dim() != 3 or ref.shape[0] != 1 or ref.shape[1:] != src.shape[1:] or src_valid_mask.shape != src.shape:
        raise RuntimeError("NCC: ref must be [1,Np,Npx], src and src_valid_mask [nv,Np,Npx]")
    _lib.require_gpu("NCC", ref, src, src_valid_mask)
    out = _NccFn.apply(_lib.f32c("NCC", ref), _lib.f32c("NCC", src), _lib.f32c("NCC", src_valid_mask))
    return out.squeeze()
