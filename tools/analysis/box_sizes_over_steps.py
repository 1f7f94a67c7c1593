#!/usr/bin/env python3
"""tools/analysis/box_sizes_over_steps.py — the triangle kernel's work over an optimisation run: pixel boxes of all (view, face) pairs
(histogram), the wave maximum of the box size (the trip count of a wave's flat coverage loop: 64 consecutive faces of one view), and the
large-box count — next to k_raster_tris' time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fmhr_amd import synthetic  # noqa: E402
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward  # noqa: E402
from fmhr_amd.optim import Adam  # noqa: E402

dev = torch.device("cuda:0")
scene = synthetic.scene_from_config("metric", 0, None)
n, H, W, V, F = scene["views"], scene["H"], scene["W"], scene["V"], scene["F"]
faces = scene["faces"].to(dev)
w2c, proj = scene["w2c"].contiguous().to(dev), scene["proj"].contiguous().to(dev)
sh = scene["sh0"].contiguous().to(dev)
vt = scene["vertices"].to(dev).contiguous()
tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
hs = HamStage2(faces, V, n, H, W, scene["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
lr = scene["lr"]
opt = Adam([{'params': delta, 'lr': lr["lr"]}, {'params': albedo, 'lr': lr["albedo_lr"]}])
a = lambda: (vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
stream = torch.cuda.current_stream()
fl = faces.long()
Fp = (F + 63) // 64 * 64
done = 0
print("steps   tris us  bins us | boxes: none     1      2-4     5-8    9-16   17-32    >32 | mean wave-max (cap 32)  waves with max > 16  waves with a >32 box")
for target in (32, 128, 320, 640, 960):
    while done < target:
        hs.step(*a(), optimizer=opt, params=(delta, albedo))
        done += 1
    reps = 9
    tm = bench.Timing(reps, stream)
    for i in range(reps):
        tm.attach(i)
        hs.step(*a())
    bench.Timing.detach()
    torch.cuda.synchronize()
    t = {k: sorted(tm.ms(i, bench.SLOTS.index(k)) for i in range(reps))[reps // 2] * 1e3 for k in ("raster_tris", "raster_resolve")}
    snap = hs.raw("snap", torch.int32, n * V * 2).view(n, V, 2).long()
    sv = snap[:, fl]                                               # [n, F, 3, 2]
    rej = (sv[..., 0] == -2147483648).any(-1)
    mn, mx = sv.min(2).values, sv.max(2).values
    pxa = torch.clamp((mn[..., 0] - 128 + 255) >> 8, min=0); pxb = torch.clamp((mx[..., 0] - 128) >> 8, max=W - 1)
    pya = torch.clamp((mn[..., 1] - 128 + 255) >> 8, min=0); pyb = torch.clamp((mx[..., 1] - 128) >> 8, max=H - 1)
    box = torch.clamp(pxb - pxa + 1, min=0) * torch.clamp(pyb - pya + 1, min=0)
    box = torch.where(rej, torch.zeros_like(box), box)
    edges = torch.tensor([0, 1, 2, 5, 9, 17, 33, 1 << 40], device=dev)
    hist = torch.histc(torch.bucketize(box.flatten(), edges, right=True).float(), bins=8, min=0.5, max=8.5).long().tolist()
    small = torch.where(box > 32, torch.zeros_like(box), box)
    pad = torch.zeros(n, Fp - F, dtype=box.dtype, device=dev)
    wmax = torch.cat([small, pad], 1).view(n, -1, 64).max(2).values
    wl = (torch.cat([box, pad], 1).view(n, -1, 64) > 32).any(2)
    print(f"{done:5d}  {t['raster_tris']:8.1f} {t['raster_resolve']:8.1f} | " + " ".join(f"{h:7d}" for h in hist[:7]) +
          f" | {float(wmax.float().mean()):10.2f} {int((wmax > 16).sum()):20d} {int(wl.sum()):20d}", flush=True)
