"""Round 5: the hostile-geometry differential test of tests/test_gpu_fuzz.py over MANY more seeds than the suite keeps (a hunt, not a test):
rasterize + antialias analysis of every soup bit-identical to the CPU checker, and the fused step under hostile cameras at the suite's bars.
    gpurun -- python tools/experiments/r05_fuzz_hunt.py [first_seed last_seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ham_oracle as O                                                # noqa: E402
import test_gpu_fuzz as Z                                             # noqa: E402
import test_gpu_ham as T                                              # noqa: E402
from fmhr_amd import dr                                               # noqa: E402
from util import sort_items                                           # noqa: E402

O.lib()
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (48, 2048)
dev = torch.device("cuda:0")
bad = 0
for seed in range(a, b):
    try:
        pos, tri, H, W = Z.hostile_soup(seed)
        r_ref, _ = O.rasterize(None, pos, tri, (H, W))
        r_gpu, _ = dr.rasterize(dr.RasterizeGLContext(), pos.to(dev), tri.to(dev), resolution=(H, W))
        assert torch.equal(r_gpu.cpu(), r_ref), "rast"
        color = torch.tensor(np.random.default_rng(seed).uniform(0, 1, (pos.shape[0], H, W, 3)), dtype=torch.float32)
        a_ref, it_ref, al_ref = O.antialias(color, r_ref, pos, tri, return_items=True)
        a_gpu, it_gpu, al_gpu, overflow = dr.antialias_debug_items(color.to(dev), r_gpu, pos.to(dev), tri.to(dev))
        si, sa = sort_items(it_gpu, al_gpu)
        assert overflow == 0 and si.shape[0] == it_ref.shape[0], "item count"
        if it_ref.shape[0]:
            assert torch.equal(si, it_ref.long()) and torch.equal(sa, al_ref), "items"
        assert float((a_gpu.cpu() - a_ref).abs().max()) < 1e-6, "aa values"
    except Exception as e:                                            # noqa: BLE001
        bad += 1
        print("SOUP FAIL seed", seed, repr(e)[:200], flush=True)
    if seed % 256 == 0:
        print("soups up to", seed, "failures", bad, flush=True)
print("soups", a, "..", b, "failures", bad, flush=True)
badc = 0
for dims in ((3, 64, 64, 24, 32), (3, 96, 80, 12, 16), (2, 128, 128, 32, 48), (5, 64, 128, 20, 28)):
    for seed in range(6, 6 + max(1, (b - a) // 64)):
        try:
            sc = Z.hostile_cameras(O, dims, seed)
            T._step_vs_oracle(O, sc, f"hunt {dims} seed {seed}")
        except Exception:                                             # noqa: BLE001
            badc += 1
            print("CAMERA FAIL", dims, seed, traceback.format_exc()[-500:], flush=True)
print("hostile cameras: failures", badc, flush=True)
