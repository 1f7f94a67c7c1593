"""fmhr_amd — MI355X-native differentiable mesh renderer for the FMHR HAM inverse-rendering loop.

Only the hot path of /root/reference/mesh_sfs_optim.py (stage-2 "HAM" iteration) lives here: HIP kernels + C ABI
(`csrc/`, `include/fmhr_ham.h`) and the host-side mirror of the reference's operator surface:

* ``fmhr_amd.dr``     — nvdiffrast-shaped ``RasterizeGLContext / rasterize / interpolate / antialias``
* ``fmhr_amd.utils``  — ``get_normals / get_radiance / get_matrix / laplacian_smoothing`` of models/utils.py
* ``fmhr_amd.ham``    — the fused stage-2 step and the optimisation loop
* ``fmhr_amd.dist``   — view sharding + RCCL all-reduce of the per-vertex gradients

There is no CPU fallback: every op raises if the HIP library is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
