"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads without a GPU and exports exactly the symbols
include/fmhr_ham.h declares; argument errors are reported without touching the GPU; the product never imports oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call([os.path.join(ROOT, "fmhr_amd", "csrc", "build.sh")])
    from fmhr_amd import _lib
    return _lib.load()


def test_header_and_library_agree(lib):
    from fmhr_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fmhr_ham.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fmhr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (fmhr_[a-z_0-9]+)", out))
    assert declared <= exported, declared - exported
    assert lib.fmhr_abi_version() == 6


def test_struct_layouts_match_header():
    from fmhr_amd._lib import HamDims, HamWeights, HamViews
    assert ctypes.sizeof(HamDims) == 40 and ctypes.sizeof(HamWeights) == 44 and ctypes.sizeof(HamViews) == 48
    from fmhr_amd._lib import AdamParam
    assert ctypes.sizeof(AdamParam) == 48
    from fmhr_amd._lib import Stage1Plan
    assert ctypes.sizeof(Stage1Plan) == 16 + 10 * 8


def test_argument_errors_without_gpu(lib):
    from fmhr_amd._lib import HamDims
    assert lib.fmhr_rasterize_fwd(None, None, 1, 3, 1, 0, 8, None, None, 0, None) == -1
    assert b"bad dims" in lib.fmhr_last_error()
    assert lib.fmhr_rasterize_fwd(None, None, 1, 3, 1, 8, 8, None, None, 0, None) == -1
    assert lib.fmhr_interpolate_fwd(None, None, None, 2, 3, 3, 1, 8, 8, 4, None, None) == -1
    assert b"attr batch" in lib.fmhr_last_error()
    assert lib.fmhr_laplacian_uniform_fwd(None, None, None, 10, 5, 1.0, None, None, None) == -1
    bad = HamDims(0, 10, 10, 8, 8, 0, 16, 0)
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(bad)) == 0
    # per-vertex tables are addressed through 32-bit byte offsets (common.h: at32): V < 2^27, said so instead of wrapping
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(HamDims(2, 1 << 27, 180, 64, 48, 0, 4096, 0))) == 0
    assert b"V < 2^27" in lib.fmhr_last_error()
    assert lib.fmhr_rasterize_fwd(None, None, 1, 1 << 27, 1, 8, 8, None, None, 0, None) == -1
    assert b"2^27" in lib.fmhr_last_error()
    ok = HamDims(2, 100, 180, 64, 48, 0, 4096, 0)
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(ok)) > 2 * 64 * 48 * 16
    # ABI 5: the row pitch — default W rounded up to a multiple of 64; W itself or a multiple of 64 above it when given
    assert lib.fmhr_ham_pitch(ctypes.byref(ok)) == 64
    assert lib.fmhr_ham_pitch(ctypes.byref(HamDims(2, 100, 180, 512, 334, 0, 4096, 0, 0))) == 384
    assert lib.fmhr_ham_pitch(ctypes.byref(HamDims(2, 100, 180, 512, 334, 0, 4096, 0, 334))) == 334
    assert lib.fmhr_ham_pitch(ctypes.byref(HamDims(2, 100, 180, 512, 334, 0, 4096, 0, 448))) == 448
    assert lib.fmhr_ham_pitch(ctypes.byref(HamDims(2, 100, 180, 512, 334, 0, 4096, 0, 400))) == -1
    assert b"pitch" in lib.fmhr_last_error()
    # ABI 6: a view selection (n_src > 0: the step's views are a subset of the capture's) needs every segment inside one view
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(HamDims(2, 100, 180, 64, 48, 0, 4096, 0, 0, 5))) > 0
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(HamDims(2, 100, 180, 50, 38, 0, 4096, 0, 38, 5))) == 0
    assert b"multiple of 64" in lib.fmhr_last_error()
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(HamDims(2, 100, 180, 64, 48, 0, 4096, 0, 0, -1))) == 0
    flat, padded = HamDims(8, 778, 1538, 512, 334, 0, 4096, 0, 334), HamDims(8, 778, 1538, 512, 334, 0, 4096, 0, 0)
    assert lib.fmhr_ham_workspace_bytes(ctypes.byref(padded)) > lib.fmhr_ham_workspace_bytes(ctypes.byref(flat)) > 0
    assert lib.fmhr_ham_rasterizer_form(ctypes.byref(HamDims(2, 49281, 98432, 512, 334, 0, 4096, 0, 0))) == 1     # W = 334, micro-triangle: bins
    assert lib.fmhr_ham_rasterizer_form(ctypes.byref(HamDims(2, 49281, 98432, 512, 334, 0, 4096, 0, 334))) == 0   # ABI-4 layout: atomic keys
    assert lib.fmhr_ham_rasterizer_form(ctypes.byref(padded)) == 0                                                 # coarse mesh: the queue
    # depth keys + the large-box counter + the large-box queue (two halves: double-buffered by the step's parity) + the snapped vertices
    assert lib.fmhr_rasterize_workspace_bytes(2, 7, 10, 8, 8) == 2 * 64 * 8 + 16 + 2 * (2 * 10 * 8) + 2 * 7 * 8
    # fmhr_ham_densify re-shades / recomputes the views the step does not store: it needs the last forward's tri and sh
    lib.fmhr_ham_densify.restype = ctypes.c_int
    lib.fmhr_ham_densify.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
    assert lib.fmhr_ham_densify(ctypes.byref(ok), None, None, None, 0, None) == -1
    assert b"ham_densify" in lib.fmhr_last_error()
    # the stage-1 entry points check their arguments before they touch the device
    from fmhr_amd._lib import Stage1Plan
    assert lib.fmhr_stage1_step(None, None, None, None, 50.0, None, None, None, None, None, None, 0, None) == -1
    assert b"plan is null" in lib.fmhr_last_error()
    plan = Stage1Plan(2, 10, 300, 0, *([None] * 10))             # rows must be a multiple of 256
    assert lib.fmhr_stage1_step(ctypes.byref(plan), None, None, None, 50.0, None, None, None, None, None, None, 0, None) == -1
    assert b"bad plan" in lib.fmhr_last_error()
    plan = Stage1Plan(2, 10, 512, 0, *([None] * 10))
    assert lib.fmhr_stage1_step(ctypes.byref(plan), None, None, None, 50.0, None, None, None, None, None, None, 0, None) == -1
    assert b"null table" in lib.fmhr_last_error()
    assert lib.fmhr_stage1_step_workspace_bytes(ctypes.byref(plan)) >= 512 * 12 + 2 * 40
    assert lib.fmhr_stage1_rows(None, None, 0, 8, 8, None, None, None, None, 0, None, 0, None) == -1
    assert lib.fmhr_stage1_transpose(None, None, None, None, 1, 10, 0, None, None, None, None, 0, None) == -1
    assert lib.fmhr_sh_normal_equations(None, None, None, None, 1, 8, 8, 4, None, None) == -1


def test_product_fails_loudly_on_cpu_tensors():
    import torch
    from fmhr_amd import dr, utils
    pos = torch.zeros(1, 3, 4)
    tri = torch.zeros(1, 3, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="GPU"):
        dr.rasterize(dr.RasterizeGLContext(), pos, tri, (8, 8))
    with pytest.raises(RuntimeError, match="GPU"):
        utils.get_normals(torch.zeros(1, 3, 3), tri)
    with pytest.raises(RuntimeError):
        dr.rasterize(None, torch.zeros(3, 4), tri, (8, 8))
    with pytest.raises(RuntimeError):
        dr.rasterize(None, pos, tri.long(), (8, 8))


def test_product_never_touches_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "fmhr_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                txt = open(os.path.join(dp, f)).read()
                # docstrings may CITE /root/reference file:line; nothing may import/open it or anything under oracle/
                if re.search(r"(import\s+ham_oracle|from\s+oracle|oracle/|raster_oracle|sys\.path[^\n]*reference|open\([^\n]*reference)", txt):
                    bad.append(f)
    assert not bad, bad
