#!/usr/bin/env python3
"""What an EMPTY segment costs the second rasterizer pass: the forward of the metric scene with its own cameras, and with the cameras
turned away (no fragment anywhere: every wave of k_raster_bins leaves after its count load).  HIP events per launch group.
    python3 tools/micro/bins_empty.py [--views 48]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from fmhr_amd import synthetic                  # noqa: E402
from fmhr_amd.ham import HamStage2, edge_length_mean   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--config", default="metric")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    scene = synthetic.scene_from_config(args.config, 0, args.views)
    n, H, W, V = scene["views"], scene["H"], scene["W"], scene["V"]
    faces = scene["faces"].to(dev)
    w2c, proj, sh = scene["w2c"].to(dev).contiguous(), scene["proj"].to(dev).contiguous(), scene["sh0"].to(dev).contiguous()
    vtmp = scene["vertices"].to(dev).contiguous()
    hs = HamStage2(faces, V, n, H, W, scene["weights"], edge_length_mean(vtmp, faces), n_global=n)
    delta = torch.zeros(V, 3, device=dev)
    albedo = scene["albedo0"].to(dev).contiguous()
    mask = torch.ones(n, H, W, device=dev)
    away = w2c.clone()
    away[:, 3, 0] += 50.0                       # row-vector convention: translation in the last ROW -> the mesh leaves every image
    stream = torch.cuda.current_stream()
    slots = [bench.SLOTS.index(s) for s in ("raster_tris", "raster_resolve")]
    for name, cam in (("own cameras", w2c), ("turned away", away), ("own cameras", w2c)):
        for _ in range(5):
            hs.forward(vtmp, delta, albedo, sh, cam, proj, mask, mask)
        K = 20
        tk = bench.Timing(K, stream)
        for i in range(K):
            tk.attach(i)
            hs.forward(vtmp, delta, albedo, sh, cam, proj, mask, mask)
            bench.Timing.detach()
        torch.cuda.synchronize()
        t = [sum(tk.ms(i, s) for i in range(K)) / K * 1e3 for s in slots]
        cov = float(hs.n_valid_local.sum()) / (n * H * W)
        print("%-12s  k_raster_tris %.1f us   k_raster_bins %.1f us   (event pair ~4.7 us included)  coverage %.3f" % (name, t[0], t[1], cov))


if __name__ == "__main__":
    main()
