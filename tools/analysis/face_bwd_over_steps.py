#!/usr/bin/env python3
"""tools/analysis/face_bwd_over_steps.py — k_face_bwd's time over an optimisation run, next to what it works on: owner (view, face)
pairs, their candidate pixels (set bits of the coverage masks), large-box entries, owned pixels, antialias items.  (Its time differed by
20 % between bench processes; the processes had stopped their pre-roll after 160 .. 320 steps.)"""
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
keys = ("shade_bwd", "raster_tris", "raster_resolve", "aa_fwd", "aa_bwd", "loss_pix")
fl = faces.long()


def stats():
    st = int(hs.raw("stamp", torch.int32, 1)) & 255
    own = hs.raw("own8", torch.uint8, n * F).view(n, F) == st
    cov = hs.raw("cov16", torch.int32, n * F * 2).view(n, F, 2)
    wfield = (cov[..., 1] >> 26) & 63
    small = own & (wfield != 63)
    bits = cov[..., 0][small]
    cand = 0
    for sft in range(0, 32, 8):
        cand += int(torch.bincount(((bits >> sft) & 255).long(), minlength=256).mul(POP).sum())
    boxw = (wfield[small] + 1).float()
    v = vt + delta.detach()
    e = torch.cat([(v[fl[:, 0]] - v[fl[:, 1]]).norm(dim=1), (v[fl[:, 1]] - v[fl[:, 2]]).norm(dim=1), (v[fl[:, 2]] - v[fl[:, 0]]).norm(dim=1)])
    lg = (own & (wfield == 63)).nonzero()
    global LARGE_BOXES
    LARGE_BOXES = None
    if lg.numel():
        snap = hs.raw("snap", torch.int32, n * V * 2).view(n, V, 2)
        b_, f_ = lg[:, 0], lg[:, 1]
        sv = torch.stack([snap[b_, fl[f_, k]] for k in range(3)], 1).long()      # [L, 3, 2]
        mn, mx = sv.min(1).values, sv.max(1).values
        pxa = torch.clamp((mn[:, 0] - 128 + 255) >> 8, min=0); pxb = torch.clamp((mx[:, 0] - 128) >> 8, max=W - 1)
        pya = torch.clamp((mn[:, 1] - 128 + 255) >> 8, min=0); pyb = torch.clamp((mx[:, 1] - 128) >> 8, max=H - 1)
        LARGE_BOXES = ((pxb - pxa + 1) * (pyb - pya + 1)).cpu()
    return dict(owners=int(own.sum()), cand=cand, large=int((own & (wfield == 63)).sum()), boxw=float(boxw.mean()), edge_cv=float(e.std() / e.mean()),
                aa=int(hs.aa_item_count()))


LARGE_BOXES = None
POP = torch.tensor([bin(i).count("1") for i in range(256)], device=dev)
done = 0
print("steps  " + "  ".join(k.rjust(14) for k in keys) + "   owners  candidates  cand/owner  large  box width  edge-length cv  aa items  n_valid")
for target in (32, 64, 128, 192, 256, 320, 448, 640, 960):
    while done < target:
        hs.step(*a(), optimizer=opt, params=(delta, albedo))
        done += 1
    reps = 9
    tm = bench.Timing(reps, stream)
    for i in range(reps):
        tm.attach(i)
        hs.step(*a())                       # no update: the same state nine times
    bench.Timing.detach()
    torch.cuda.synchronize()
    t = {k: sorted(tm.ms(i, bench.SLOTS.index(k)) for i in range(reps))[reps // 2] * 1e3 for k in keys}
    s = stats()
    print(f"{done:5d}  " + "  ".join(f"{t[k]:14.1f}" for k in keys) + f"   {s['owners']:7d}  {s['cand']:10d}  {s['cand'] / max(s['owners'], 1):10.2f}  {s['large']:5d}  "
          f"{s['boxw']:9.2f}  {s['edge_cv']:14.3f}  {s['aa']:8d}  {int(hs.losses[7])}", flush=True)
    if LARGE_BOXES is not None:
        q = LARGE_BOXES.float()
        print("        large boxes (pixels): n %d  median %.0f  p90 %.0f  max %.0f  | <= 48: %d  <= 64: %d  <= 128: %d" % (
            q.numel(), q.median(), q.quantile(0.9), q.max(), int((q <= 48).sum()), int((q <= 64).sum()), int((q <= 128).sum())), flush=True)
