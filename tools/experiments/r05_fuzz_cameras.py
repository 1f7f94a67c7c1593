"""Round 5 exploration: the fused step under hostile cameras (off-screen, zoomed in / out, mirrored, camera inside the mesh) against the
checker — prints what the bars of tests/test_gpu_ham.py::_step_vs_oracle would say instead of asserting.
    gpurun -- python tools/experiments/r05_fuzz_cameras.py"""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ham_oracle as O                                                # noqa: E402
import test_gpu_ham as T                                              # noqa: E402
import test_gpu_fuzz as Z                                             # noqa: E402

O.lib()
for dims in ((3, 64, 64, 24, 32), (3, 96, 80, 12, 16), (4, 128, 128, 32, 48)):
    for seed in range(6):
        try:
            sc = Z.hostile_cameras(O, dims, seed)
            ld, v = T._step_vs_oracle(O, sc, f"{dims} seed {seed}")
            print("ok", dims, seed, "binned", v["binned"], "coverage", float((v["rast"][..., 3] > 0).float().mean()), flush=True)
        except Exception:
            print("FAIL", dims, seed, traceback.format_exc()[-600:], flush=True)
