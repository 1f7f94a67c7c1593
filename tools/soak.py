#!/usr/bin/env python3
"""tools/soak.py — a long optimisation run on ONE workspace (default 3000 steps of the metric workload: five times the reference's demo
run), then the step at the final parameters against the same step on a FRESH workspace: everything the step keeps between calls (bin
cursors, owner marks, coverage entries, the large-box queue and its per-vertex buffer, antialias lists) must have cleaned itself, with
thousands of sliver triangles in play; overflow flags are checked.  usage: soak.py [steps] [config]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic  # noqa: E402
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward  # noqa: E402
from fmhr_amd.optim import Adam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = sys.argv[2] if len(sys.argv) > 2 else "metric"
dev = torch.device("cuda:0")
scene = synthetic.scene_from_config(cfg, 0, None)
n, H, W, V, F = scene["views"], scene["H"], scene["W"], scene["V"], scene["F"]
faces = scene["faces"].to(dev)
w2c, proj = scene["w2c"].contiguous().to(dev), scene["proj"].contiguous().to(dev)
sh = scene["sh0"].contiguous().to(dev)
vt = scene["vertices"].to(dev).contiguous()
tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
elm = edge_length_mean(vt, faces)
hs = HamStage2(faces, V, n, H, W, scene["weights"], elm)
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': scene["lr"]["lr"]}, {'params': albedo, 'lr': scene["lr"]["albedo_lr"]}])
rest = (sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
first = None
for it in range(steps):
    losses, _, _ = hs.step(vt, delta.detach(), albedo.detach()[0], *rest, optimizer=opt, params=(delta, albedo))
    if it == 0:
        first = losses.clone()
    if (it + 1) % 500 == 0:
        ld = hs.loss_dict()                          # raises on an antialias-list / fragment-bin overflow
        print(f"step {it + 1}: total {ld['total']:.4f} (first {float(first[6]):.4f})  n_valid {int(ld['n_valid'])}  aa items {int(hs.aa_item_count())}", flush=True)
d_fin, a_fin = delta.detach().clone(), albedo.detach()[0].clone()
hs.step(vt, d_fin, a_fin, *rest)
old = [t.clone() for t in (hs.losses, hs.gdelta, hs.galbedo)]
st = int(hs.raw("stamp", torch.int32, 1)) & 255
own = hs.raw("own8", torch.uint8, n * F).view(n, F) == st
cov = hs.raw("cov16", torch.int32, n * F * 2).view(n, F, 2)
large = int((own & (((cov[..., 1] >> 26) & 63) == 63)).sum())
fresh = HamStage2(faces, V, n, H, W, scene["weights"], elm)
fresh.step(vt, d_fin, a_fin, *rest)
new = [fresh.losses, fresh.gdelta, fresh.galbedo]
rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
e = [rel(x, y) for x, y in zip(old, new)]
print(f"after {steps} steps ({cfg}): {large} large-box (view, face)s among the owners; step on the old workspace vs a fresh one: "
      f"losses {e[0]:.1e} gdelta {e[1]:.1e} galbedo {e[2]:.1e}")
assert torch.isfinite(old[0]).all() and max(e) < 1e-5, e
print("soak ok")
