#!/usr/bin/env python3
"""tools/graph_probe.py — is a hipGraph replay of the fused step shorter than its ten eager launches?  The step (forward + backward,
no optimiser update: Adam's step count is a host argument) is captured once with torch.cuda.CUDAGraph and replayed; both forms are
timed with HIP events over the same parameters.  usage: graph_probe.py [views ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic  # noqa: E402
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward  # noqa: E402
from fmhr_amd.optim import Adam  # noqa: E402


def run(views, steps=200):
    dev = torch.device("cuda:0")
    scene = synthetic.scene_from_config("metric", 0, views)
    n, H, W, V = scene["views"], scene["H"], scene["W"], scene["V"]
    faces = scene["faces"].to(dev)
    w2c, proj = scene["w2c"].contiguous().to(dev), scene["proj"].contiguous().to(dev)
    sh = scene["sh0"].contiguous().to(dev)
    vt = scene["vertices"].to(dev).contiguous()
    tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    hs = HamStage2(faces, V, n, H, W, scene["weights"], edge_length_mean(vt, faces))
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': scene["lr"]["lr"]}, {'params': albedo, 'lr': scene["lr"]["albedo_lr"]}])
    a = (vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
    for _ in range(160):                       # to the stationary workload (bench.py's pre-roll)
        hs.step(*a, optimizer=opt, params=(delta, albedo))
    torch.cuda.synchronize()

    def timed(fn):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eager_side = timed(lambda: hs.step(*a))
    eager = timed(lambda: hs.step(*a))
    ref = [t.clone() for t in hs.step(*a)]
    torch.cuda.synchronize()                   # the next step runs on another (non-blocking) stream: it must not overlap this one
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        hs.step(*a)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            hs.step(*a)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    out = [hs.losses.clone(), hs.gdelta.clone(), hs.galbedo.clone()]
    same = all(torch.equal(x, y) for x, y in zip(ref, out))
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
    err = max(rel(x, y) for x, y in zip(out, ref))
    if os.environ.get("PROBE_DIAG"):
        print("   replay vs eager  : losses %.1e gdelta %.1e galbedo %.1e" % tuple(rel(x, y) for x, y in zip(out, ref)), hs.losses.tolist())
        for k in range(3):
            g.replay(); torch.cuda.synchronize()
            o2 = [hs.losses.clone(), hs.gdelta.clone(), hs.galbedo.clone()]
            print("   replay %d vs eager: losses %.1e gdelta %.1e galbedo %.1e" % ((k + 2,) + tuple(rel(x, y) for x, y in zip(o2, ref))), "stamp", int(hs.raw("stamp", torch.int32, 1)))
        o3 = [t.clone() for t in hs.step(*a)]; torch.cuda.synchronize()
        print("   eager again      : losses %.1e gdelta %.1e galbedo %.1e" % tuple(rel(x, y) for x, y in zip(o3, ref)), ref[0].tolist())
    graph = timed(g.replay)
    print(f"views {n:3d}: eager {eager * 1e3:7.1f} us (side stream {eager_side * 1e3:7.1f})   graph replay {graph * 1e3:7.1f} us   "
          f"replay == eager step: {'bit-identical' if same else 'max rel diff %.1e' % err}", flush=True)


if __name__ == "__main__":
    for v in [int(x) for x in sys.argv[1:]] or [6, 48]:
        run(v)
