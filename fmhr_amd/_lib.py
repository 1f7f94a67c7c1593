"""ctypes binding of libfmhr_ham.so (C ABI: include/fmhr_ham.h).  No torch types cross the boundary: raw device pointers,
sizes and the HIP stream handle only.  There is no fallback: a missing library or a failing call raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfmhr_ham.so")

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p


class HamDims(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("V", ctypes.c_int32), ("F", ctypes.c_int32), ("H", ctypes.c_int32),
                ("W", ctypes.c_int32), ("nnz", ctypes.c_int32), ("aa_cap", ctypes.c_int32), ("bin_cap", ctypes.c_int32),
                ("pitch", ctypes.c_int32), ("n_src", ctypes.c_int32)]


class HamWeights(ctypes.Structure):
    _fields_ = [("sfs", c_float), ("lap", c_float), ("albedo", c_float), ("mask", c_float), ("edge", c_float),
                ("delta", c_float), ("edge_length_mean", c_float), ("mse_denominator", c_float),
                ("reg_grad_scale", c_float), ("reg_v0", ctypes.c_int32), ("reg_v1", ctypes.c_int32)]


class HamViews(ctypes.Structure):
    _fields_ = [("pos", c_void_p), ("normals", c_void_p), ("rast", c_void_p), ("shaded", c_void_p), ("aa", c_void_p),
                ("aa_counter", c_void_p)]


class AdamParam(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("lr", c_float), ("beta1", c_float),
                ("beta2", c_float), ("eps", c_float), ("step", ctypes.c_int32)]


class Stage1Plan(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("V", ctypes.c_int32), ("rows", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("view_row0", c_void_p), ("view_valid", c_void_p), ("row_ptr", c_void_p), ("ent_col", c_void_p), ("ent_w", c_void_p),
                ("row_n", c_void_p), ("row_img", c_void_p), ("t_ptr", c_void_p), ("t_row", c_void_p), ("t_w", c_void_p)]


P = c_void_p
# name -> (restype, argtypes); every name declared in include/fmhr_ham.h is listed (tests/test_abi.py checks both ways)
SIGNATURES = {
    "fmhr_abi_version": (c_int, []),
    "fmhr_last_error": (ctypes.c_char_p, []),
    "fmhr_topology_workspace_bytes": (c_size_t, [c_int]),
    "fmhr_topology_build": (c_int, [P, c_int, c_int, P, P, c_size_t, P]),
    "fmhr_adjacency_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fmhr_adjacency_build": (c_int, [P, c_int, c_int, P, P, P, c_size_t, P]),
    "fmhr_rasterize_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fmhr_rasterize_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "fmhr_rasterize_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "fmhr_interpolate_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "fmhr_interpolate_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "fmhr_antialias_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "fmhr_antialias_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "fmhr_vertex_normals_fwd": (c_int, [P, P, c_int, c_int, c_int, P, P, P]),
    "fmhr_vertex_normals_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fmhr_sh_radiance_fwd": (c_int, [P, P, c_int, c_int, P, P]),
    "fmhr_sh_radiance_bwd": (c_int, [P, P, P, c_int, c_int, P, P, P]),
    "fmhr_laplacian_uniform_fwd": (c_int, [P, P, P, c_int, c_int, c_float, P, P, P]),
    "fmhr_laplacian_uniform_bwd": (c_int, [P, P, P, c_int, c_int, c_float, P, P, P]),
    "fmhr_timing_attach": (c_int, [P, c_int, c_int]),
    "fmhr_timing_detach": (c_int, []),
    "fmhr_ncc_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "fmhr_ncc_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fmhr_adam_step": (c_int, [P, P, P, P, ctypes.c_longlong, c_float, c_float, c_float, c_float, c_int, P]),
    "fmhr_adam_step2": (c_int, [P, P, P, P, ctypes.c_longlong, c_float, c_float, c_float, c_float, c_int,
                                P, P, P, P, ctypes.c_longlong, c_float, c_float, c_float, c_float, c_int, P]),
    "fmhr_ham_workspace_bytes": (c_size_t, [ctypes.POINTER(HamDims)]),
    "fmhr_ham_rasterizer_form": (c_int, [ctypes.POINTER(HamDims)]),
    "fmhr_ham_pitch": (c_int, [ctypes.POINTER(HamDims)]),
    "fmhr_ham_workspace_offset": (ctypes.c_longlong, [ctypes.POINTER(HamDims), ctypes.c_char_p]),
    "fmhr_ham_workspace_views": (c_int, [ctypes.POINTER(HamDims), P, c_size_t, ctypes.POINTER(HamViews)]),
    "fmhr_ham_forward": (c_int, [ctypes.POINTER(HamDims), P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P, c_size_t, P]),
    "fmhr_ham_workspace_init": (c_int, [ctypes.POINTER(HamDims), P, c_size_t, P]),
    "fmhr_ham_densify": (c_int, [ctypes.POINTER(HamDims), P, P, P, c_size_t, P]),
    "fmhr_ham_backward": (c_int, [ctypes.POINTER(HamDims), ctypes.POINTER(HamWeights), P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                  P, P, P, P, P, P, P, P, P, ctypes.POINTER(AdamParam), ctypes.POINTER(AdamParam), P, c_size_t, P]),
    "fmhr_ham_combine": (c_int, [c_int, P, ctypes.POINTER(AdamParam), ctypes.POINTER(AdamParam), P]),
    "fmhr_edge_multiplicity_build": (c_int, [P, c_int, c_int, P, P, c_int, P, P]),
    "fmhr_corner_slots_build": (c_int, [P, c_int, P, P]),
    "fmhr_vertex_faces_workspace_bytes": (c_size_t, [c_int]),
    "fmhr_vertex_faces_build": (c_int, [P, c_int, c_int, P, P, P, c_size_t, P]),
    "fmhr_stage1_rows_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fmhr_stage1_rows": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, ctypes.c_longlong, P, c_size_t, P]),
    "fmhr_stage1_row_ptr_workspace_bytes": (c_size_t, [c_int]),
    "fmhr_stage1_row_ptr": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "fmhr_stage1_fill": (c_int, [P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "fmhr_stage1_transpose_workspace_bytes": (c_size_t, [c_int, c_int]),
    "fmhr_stage1_transpose": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "fmhr_sh_normal_equations": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "fmhr_sh_albedo_sums": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "fmhr_peer_handle_bytes": (c_size_t, []),
    "fmhr_peer_create": (c_int, [c_int, c_int, ctypes.c_longlong, ctypes.POINTER(c_void_p)]),
    "fmhr_peer_export": (c_int, [P, P]),
    "fmhr_peer_import": (c_int, [P, c_int, P]),
    "fmhr_peer_all_reduce": (c_int, [P, P, ctypes.c_longlong, P]),
    "fmhr_peer_staging": (c_int, [P, ctypes.POINTER(P), ctypes.POINTER(ctypes.c_longlong)]),
    "fmhr_peer_status": (c_int, [P, ctypes.POINTER(c_int)]),
    "fmhr_peer_destroy": (c_int, [P]),
    "fmhr_stage1_step_workspace_bytes": (c_size_t, [ctypes.POINTER(Stage1Plan)]),
    "fmhr_stage1_step": (c_int, [ctypes.POINTER(Stage1Plan), P, P, P, c_float, P, P, P, ctypes.POINTER(AdamParam),
                                 ctypes.POINTER(AdamParam), P, c_size_t, P]),
}

_lib = None


def load():
    """Load libfmhr_ham.so (built by __graft_entry__.build() / fmhr_amd/csrc/build.sh).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"fmhr_amd: {LIB_PATH} is missing — build it with fmhr_amd/csrc/build.sh "
                               "(there is no CPU fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.fmhr_abi_version() != 6:
            raise RuntimeError("fmhr_amd: ABI version mismatch")
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.fmhr_last_error().decode()}")


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_gpu(name, *tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name}: tensors must live on the GPU (fmhr_amd has no CPU path)")


def f32c(name, t):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()
