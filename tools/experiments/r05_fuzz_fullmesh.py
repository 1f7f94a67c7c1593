"""Round 5 hunt: hostile cameras (tests/test_gpu_fuzz.py) on the EXACT sub3 mesh (98 432 faces) — zoomed out, the whole mesh lands in a
handful of 64-pixel segments (fragment bins far beyond their default capacity: the overflow must be REPORTED, and a workspace sized from the
reported depth must reproduce the checker); zoomed in, most triangles are large boxes (queue, sliver workers, in-wave sweeps).
    gpurun -- python tools/experiments/r05_fuzz_fullmesh.py [seeds]"""
import os
import sys

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
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
bad = 0
for dims in ((2, 192, 192, "mano", 0, 3), (3, 128, 256, "mano", 0, 3)):
    for s in range(seeds):
        sc = Z.hostile_cameras(O, dims, 40 + s)
        g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
        cap, tries = 0, 0
        while True:
            hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], bin_cap=cap)
            _, gd, ga = hs.step(g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
            try:
                ld = hs.loss_dict()
                break
            except RuntimeError as e:
                need = hs.bin_depth_max()
                print("  overflow reported:", str(e)[:150], "-> bin_cap", need, flush=True)
                assert "fragment bin" in str(e) and need > (cap or 1024) and tries < 2
                cap, tries = need, tries + 1
        v = hs.views()
        d = sc["delta"].clone().requires_grad_(True)
        a = sc["albedo"][None].clone().requires_grad_(True)
        ref = O.ham_stage2_losses(sc["vertices"], d, a, sc["sh0"], sc["w2c"], sc["proj"], sc["img"], sc["mask"], sc["valid_mask"], sc["faces"],
                                  (sc["H"], sc["W"]), sc["weights"], hs.edge_len_mean, pos_value=v["pos"].cpu())
        ref["loss"].backward()
        ok = torch.equal(v["rast"].cpu(), ref["rast"])
        el = max(abs(ld[k] - float(ref[k])) / max(abs(float(ref[k])), 1e-6) for k in ("sfs", "lap", "albedo", "mask", "edge", "delta"))
        eg = (rel_err(gd, d.grad), rel_err(ga, a.grad[0]))
        if not ok or el > 1e-3 or max(eg) > 1e-3:
            bad += 1
            print("FAIL", dims, s, "rast", ok, el, eg, flush=True)
        print(dims, "seed", s, "binned", hs.binned, "fullest bin", hs.bin_depth_max(), "cap", cap or 1024, "coverage",
              round(float((v["rast"][..., 3] > 0).float().mean()), 3), "failures", bad, flush=True)
print("full-mesh hunt: failures", bad, flush=True)
