"""Round 5 hunt: the fused stage-1 iteration (HamStage1) under hostile cameras (tests/test_gpu_fuzz.py: hostile_cameras) against the loop body
of mesh_sfs_optim.py:205-236 run op by op on the product's generic ops with autograd — loss, albedo gradient, SH gradient, whole batches and
random subsets.    gpurun -- python tools/experiments/r05_fuzz_stage1.py [seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import ham_oracle as O                                                # noqa: E402
import test_gpu_fuzz as Z                                             # noqa: E402
from test_gpu_stage1 import _reference                                # noqa: E402
from fmhr_amd import pipeline                                         # noqa: E402
from fmhr_amd.stage1 import HamStage1                                 # noqa: E402
from util import rel_err                                              # noqa: E402

O.lib()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops = pipeline.product_ops()
bad = 0
for dims in ((4, 64, 64, 24, 32), (3, 96, 80, 12, 16), (5, 48, 334, 16, 24), (3, 128, 128, 32, 48)):
    for s in range(seeds):
        sc = Z.hostile_cameras(O, dims, 900 + s)
        res = (sc["H"], sc["W"])
        g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "faces", "img", "mask", "w2c", "proj")}
        torch.manual_seed(s)
        albedo = (sc["albedo"][None].to(dev) + 0.05 * torch.randn(1, sc["V"], 3, device=dev)).contiguous()
        sh = (sc["sh0"].to(dev) + 0.05 * torch.randn(sc["views"], 9, device=dev)).contiguous()
        hs = HamStage1(g["vertices"], g["faces"], g["w2c"], g["proj"], g["img"], g["mask"], res, 50.0)
        rng = np.random.default_rng(s)
        for idx in (None, torch.tensor(rng.permutation(sc["views"])[:2].copy(), device=dev)):
            lv, ga, gs = hs.step(albedo, sh, idx=idx)
            lr, gar, gsr = _reference(pipeline, ops, g["vertices"], g["faces"], albedo, sh, g["w2c"], g["proj"], g["img"], g["mask"], res, idx=idx)
            lf, lrf = float(lv.sum()), float(lr)
            if not np.isfinite(lrf):                                  # a batch without a valid pixel: torch's mean over nothing is NaN, the fused step says 0
                ok = lf == 0.0 and float(ga.abs().max()) == 0.0
            else:
                ok = abs(lf - lrf) <= 1e-4 * max(abs(lrf), 1e-6) and rel_err(ga, gar) < 1e-4 and rel_err(gs, gsr) < 1e-4
            if not ok:
                bad += 1
                print("FAIL", dims, s, None if idx is None else idx.tolist(), lf, lrf, rel_err(ga, gar), rel_err(gs, gsr), flush=True)
    print(dims, "failures so far", bad, flush=True)
print("stage-1 hunt: failures", bad, flush=True)
