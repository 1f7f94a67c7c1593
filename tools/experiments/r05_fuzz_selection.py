"""Round 5 hunt: the minibatch selection inside the step (ABI 6) on ONE workspace over many steps — a capture of `num` hostile views
(tests/test_gpu_fuzz.py: hostile_cameras), a random subset of n of them every step (order included), against the checker on the
GATHERED inputs: `rast` bit-identical, losses and gradients of its fp32 evaluation.  The per-source-view valid-mask table, the state
words and the lists must follow whatever subset comes.
    gpurun -- python tools/experiments/r05_fuzz_selection.py [scenes steps]"""
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
scenes, steps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 10)
dev = torch.device("cuda:0")
bad = 0
for dims, n in (((7, 64, 64, 24, 32), 3), ((6, 96, 128, 20, 28), 4), ((5, 64, 192, 40, 56), 2), ((8, 48, 334, 16, 24), 5)):
    for s in range(scenes):
        sc = Z.hostile_cameras(O, dims, 300 + s)
        num = sc["views"]
        rng = np.random.default_rng(s)
        hs = HamStage2(sc["faces"].to(dev), sc["V"], n, sc["H"], sc["W"], sc["weights"], sc["elm"], n_src=num)
        g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
        for t in range(steps):
            idx = torch.tensor(rng.permutation(num)[:n].copy(), dtype=torch.int64)
            delta = torch.tensor(rng.standard_normal((sc["V"], 3)) * 3e-4, dtype=torch.float32)      # the mesh moves between steps
            _, gd, ga = hs.step(g["vertices"], delta.to(dev), g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"],
                                view_idx=idx.to(torch.int32).to(dev))
            ld, v = hs.loss_dict(), hs.views()
            d = delta.clone().requires_grad_(True)
            a = sc["albedo"][None].clone().requires_grad_(True)
            ref = O.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"][idx], sc["w2c"][idx], sc["proj"][idx], sc["img"][idx], sc["mask"][idx],
                                      sc["valid_mask"][idx], sc["faces"], (sc["H"], sc["W"]), sc["weights"], hs.edge_len_mean, pos_value=v["pos"].cpu())
            ref["loss"].backward()
            ok = torch.equal(v["rast"].cpu(), ref["rast"])
            el = max(abs(ld[k] - float(ref[k])) / max(abs(float(ref[k])), 1e-6) for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"))
            eg = (rel_err(gd, d.grad), rel_err(ga, a.grad[0]))
            if not ok or el > 1e-3 or max(eg) > 1e-3:
                bad += 1
                print("FAIL", dims, "scene", s, "step", t, idx.tolist(), "rast", ok, "losses", el, "grads", eg, flush=True)
        print(dims, n, "of", num, "scene", s, "failures so far", bad, "binned", hs.binned, flush=True)
print("selection hunt: failures", bad, flush=True)
