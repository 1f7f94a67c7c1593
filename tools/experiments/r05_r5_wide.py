"""Round 5, CPU experiment (no GPU): what rule R5's VALUES cost in fp32, and what evaluating them in double buys.

The two scenes of RELAXED in tests/test_gpu_ham.py (round 4) — the reference capture's own cameras at 1280 x 1024 and the two-hand
sub3 mesh at 512 x 334 — through the oracle: the fp32 twin's gradients against "fp64 values on the fp32 decisions", once with the
all-fp32 rule R5 (orc_set_r5_wide(0): round 4) and once with the barycentrics and their position gradient in double (round 5).

    python tools/experiments/r05_r5_wide.py [capture|two_hands|micro] ...
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ham_oracle as oracle                      # noqa: E402
from util import rel_err, mesh_scene             # noqa: E402
import test_gpu_ham as T                         # noqa: E402


def scene(name):
    from fmhr_amd import synthetic as S
    if name == "capture":
        sc, g, cams, X = T.demo_capture_scene()
        tg = S.make_targets(sc, T.oracle_forward_fn(oracle, sc))
        gen = torch.Generator().manual_seed(14)
        sc["delta"] = (2e-4 * sc["scale"]) * torch.randn(sc["V"], 3, generator=gen)
        sc["albedo"] = sc["albedo0"] + 0.05 * torch.rand(sc["V"], 3, generator=gen)
        sc.update({k: tg[k] for k in ("img", "mask", "valid_mask")})
        sc["elm"] = float(oracle.edge_length_mean(sc["vertices"], sc["faces"]))
        return sc
    dims, seed = {"two_hands": ((2, 512, 334, "mano", 0, 3, 2), 11), "micro": ((2, 192, 192, 64, 96), 2),
                  "metric3": ((3, 512, 512, "mano", 0, 3), 12)}[name]
    return T.prepare(oracle, dims, seed)


def grads(sc, dtype, pos_value, decisions=None):
    c = lambda t: t.to(dtype)
    d = c(sc["delta"]).clone().requires_grad_(True)
    a = c(sc["albedo"])[None].clone().requires_grad_(True)
    ref = oracle.ham_stage2_losses(c(sc["vertices"]), d, a, c(sc["sh0"]), c(sc["w2c"]), c(sc["proj"]), c(sc["img"]), c(sc["mask"]),
                                   c(sc["valid_mask"]), sc["faces"], (sc["H"], sc["W"]), sc["weights"], sc["elm"], pos_value=pos_value,
                                   decisions=decisions)
    ref["loss"].backward()
    return ref, d.grad, a.grad[0]


def main():
    names = sys.argv[1:] or ["micro", "two_hands", "capture"]
    lib = oracle.lib()
    for name in names:
        lib.orc_set_r5_wide(1)
        t0 = time.time()
        sc = scene(name)
        pos32 = oracle.clip_positions(sc["vertices"] + sc["delta"], sc["w2c"], sc["proj"])[1].contiguous()
        out = {}
        for wide in (0, 1):
            lib.orc_set_r5_wide(wide)
            ref32, gd32, ga32 = grads(sc, torch.float32, pos32)
            with oracle.f32_decisions():
                ref64, gd64, ga64 = grads(sc, torch.float64, pos32, ref32["decisions"])
            out[wide] = (ref32, gd32, ga32, gd64, ga64)
            print(f"{name}: r5_wide={wide}: fp32 twin vs fp64-on-fp32-decisions: gdelta {rel_err(gd32, gd64):.2e} galbedo {rel_err(ga32, ga64):.2e}"
                  f" | rast u,v max diff vs fp64 {float((ref32['rast'][..., :2].double() - ref64['rast'][..., :2]).abs().max()):.2e}"
                  f" | ids equal {bool(torch.equal(ref32['rast'][..., 3].double(), ref64['rast'][..., 3]))}", flush=True)
        same_ids = torch.equal(out[0][0]["rast"][..., 2:], out[1][0]["rast"][..., 2:])
        print(f"{name}: z/w and ids identical between the two rules: {same_ids}; fp64 truth moved by "
              f"{rel_err(out[0][3], out[1][3]):.1e} / {rel_err(out[0][4], out[1][4]):.1e}; {time.time() - t0:.0f} s", flush=True)
    lib.orc_set_r5_wide(1)


if __name__ == "__main__":
    main()
