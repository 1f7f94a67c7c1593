"""Round 5 hunt: the op-by-op surface's BACKWARD passes (rasterize -> interpolate -> antialias, gradients with respect to the clip positions
and the attributes) on hostile triangle soups (tests/test_gpu_fuzz.py: hostile_soup) against the CPU checker's autograd, fp32 twin.
Non-finite inputs are left out here (a NaN coordinate has no gradient to compare); everything else of the generator stays.
    gpurun -- python tools/experiments/r05_fuzz_backward.py [first last]"""
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
from fmhr_amd import dr                                               # noqa: E402
from util import rel_err                                              # noqa: E402

O.lib()
a0, b0 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 600)
dev = torch.device("cuda:0")
bad = skipped = 0
worst = 0.0
for seed in range(a0, b0):
    pos, tri, H, W = Z.hostile_soup(seed)
    if not bool(torch.isfinite(pos).all()):
        skipped += 1
        continue
    rng = np.random.default_rng(seed)
    n, V, C = pos.shape[0], pos.shape[1], 4
    attr0 = torch.tensor(rng.standard_normal((n, V, C)), dtype=torch.float32)
    wt = torch.tensor(rng.standard_normal((n, H, W, C)), dtype=torch.float32)

    def run(mod, p, a, w_, tri_, ctx):
        p = p.clone().requires_grad_(True)
        a = a.clone().requires_grad_(True)
        rast, _ = mod.rasterize(ctx, p, tri_, resolution=(H, W))
        feat, _ = mod.interpolate(a, rast, tri_)
        aa = mod.antialias(feat, rast, p, tri_)
        loss = (aa * w_).sum() + (feat * w_).sum() * 0.5
        loss.backward()
        return loss.detach(), p.grad, a.grad
    l_ref, gp_ref, ga_ref = run(O, pos, attr0, wt, tri, None)
    l_gpu, gp_gpu, ga_gpu = run(dr, pos.to(dev), attr0.to(dev), wt.to(dev), tri.to(dev), dr.RasterizeGLContext())
    fin = bool(torch.isfinite(gp_ref).all()) and bool(torch.isfinite(ga_ref).all())
    if not fin:                                                       # the checker's own gradient overflows (a vertex at w = 1e-7): both must agree on that
        ok = bool(torch.isfinite(gp_gpu).all()) == bool(torch.isfinite(gp_ref).all())
        e = (0.0, 0.0)
    else:
        e = (rel_err(gp_gpu, gp_ref) if float(gp_ref.abs().max()) > 0 else float(gp_gpu.abs().max()),
             rel_err(ga_gpu, ga_ref) if float(ga_ref.abs().max()) > 0 else float(ga_gpu.abs().max()))
        ok = max(e) < 1e-4 and bool(torch.isfinite(gp_gpu).all())
        worst = max(worst, max(e))
    if not ok:
        bad += 1
        print("FAIL seed", seed, tuple(pos.shape), H, W, "finite ref", fin, "errs", e, flush=True)
    if seed % 100 == 0:
        print("up to", seed, "failures", bad, "skipped (non-finite input)", skipped, "worst", worst, flush=True)
print("backward hunt", a0, "..", b0, "failures", bad, "skipped", skipped, "worst rel err", worst, flush=True)
