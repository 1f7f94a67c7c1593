#!/usr/bin/env python3
"""tools/placement_probe.py — does k_face_bwd's time follow the WORKSPACE (its addresses) or the PROCESS?  Several HamStage2 workspaces are
built in one process (each behind a differently sized dummy allocation, all kept alive), stepped to the same state with the same inputs,
and the launch groups are timed on each with the bench's HIP-event hook; the first one is timed again at the end."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fmhr_amd import synthetic  # noqa: E402
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward  # noqa: E402

dev = torch.device("cuda:0")
scene = synthetic.scene_from_config("metric", 0, None)
n, H, W, V = scene["views"], scene["H"], scene["W"], scene["V"]
faces = scene["faces"].to(dev)
w2c, proj = scene["w2c"].contiguous().to(dev), scene["proj"].contiguous().to(dev)
sh = scene["sh0"].contiguous().to(dev)
vt = scene["vertices"].to(dev).contiguous()
tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
elm = edge_length_mean(vt, faces)
torch.manual_seed(0)
delta = (torch.randn(V, 3, device=dev) * 2e-4).contiguous()          # a roughened contour: ~the stationary workload, without an optimiser
albedo = scene["albedo0"].to(dev).contiguous()
a = (vt, delta, albedo, sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
stream = torch.cuda.current_stream()
keys = ("shade_bwd", "raster_tris", "raster_resolve", "loss_pix", "aa_fwd", "regularisers")


def measure(hs, reps=15):
    for _ in range(5):
        hs.step(*a)
    tm = bench.Timing(reps, stream)
    for i in range(reps):
        tm.attach(i)
        hs.step(*a)
    bench.Timing.detach()
    torch.cuda.synchronize()
    return {k: sorted(tm.ms(i, bench.SLOTS.index(k)) for i in range(reps))[reps // 2] * 1e3 for k in keys}


keep, objs = [], []
for k, dummy_mb in enumerate((0, 3, 64, 1000, 2048 + 1, 7)):
    if dummy_mb:
        keep.append(torch.empty(dummy_mb << 20, dtype=torch.uint8, device=dev))
    hs = HamStage2(faces, V, n, H, W, scene["weights"], elm)
    objs.append(hs)
    m = measure(hs)
    print(f"workspace {k}: base 0x{hs.ws.data_ptr():x} (mod 2 MiB: {hs.ws.data_ptr() % (2 << 20):#x}), {hs.ws_bytes / 2**30:.2f} GiB | " +
          "  ".join(f"{kk} {m[kk]:.1f}" for kk in keys), flush=True)
for k in (0, 1, 0):
    m = measure(objs[k])
    print(f"workspace {k} again: " + "  ".join(f"{kk} {m[kk]:.1f}" for kk in keys), flush=True)
