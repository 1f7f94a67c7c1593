"""Round 5 hunt: ONE workspace, a different set of hostile cameras every step (tests/test_gpu_fuzz.py: hostile_cameras) — coverage
jumps between steps, views go empty and come back, triangles switch between the small-box and the large-box path: everything the step keeps
BETWEEN steps (segment state words, bin cursors, owner marks, coverage entries, the sliver workers' queue, the antialias lists, the depth
mark) must follow.  Every step: `rast` bit-identical to the fp32 checker evaluated at the step's own positions, the six losses and both
gradients against the checker's fp32 evaluation (1e-3: a stale entry shows as O(1)).
    gpurun -- python tools/experiments/r05_fuzz_stateful.py [scenes steps]"""
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
from fmhr_amd.ham import HamStage2                                    # noqa: E402
from util import rel_err                                              # noqa: E402

O.lib()
scenes, steps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 8)
dev = torch.device("cuda:0")
bad = 0
for dims in ((3, 64, 64, 24, 32), (3, 96, 80, 12, 16), (4, 128, 128, 32, 48), (2, 64, 192, 40, 56)):
    for s in range(scenes):
        hs = None
        for t in range(steps):
            sc = Z.hostile_cameras(O, dims, 1000 * s + t)             # the same mesh family (seeded by dims), new cameras, targets, delta
            if hs is None:
                base = sc
                hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"])
            else:                                                     # keep the FIRST scene's topology: only per-step inputs change
                if sc["V"] != base["V"] or not torch.equal(sc["faces"], base["faces"]):
                    sc = dict(sc, faces=base["faces"])
            g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
            losses, gd, ga = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
            ld = hs.loss_dict()
            v = hs.views()
            d = sc["delta"].clone().requires_grad_(True)
            a = sc["albedo"][None].clone().requires_grad_(True)
            ref = O.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"], sc["w2c"], sc["proj"], sc["img"], sc["mask"], sc["valid_mask"], sc["faces"],
                                      (sc["H"], sc["W"]), sc["weights"], hs.edge_len_mean, pos_value=v["pos"].cpu())
            ref["loss"].backward()
            ok = torch.equal(v["rast"].cpu(), ref["rast"])
            el = max(abs(ld[k] - float(ref[k])) / max(abs(float(ref[k])), 1e-6) for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"))
            eg = (rel_err(gd, d.grad), rel_err(ga, a.grad[0]))
            if not ok or el > 1e-3 or max(eg) > 1e-3 or not all(np.isfinite(x) for x in ld.values()):
                bad += 1
                print("FAIL", dims, "scene", s, "step", t, "rast", ok, "losses", el, "grads", eg, flush=True)
        print(dims, "scene", s, "ok so far, failures", bad, "fullest bin", hs.bin_depth_max(), flush=True)
print("stateful hunt: failures", bad, flush=True)
