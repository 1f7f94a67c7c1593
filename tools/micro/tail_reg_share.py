#!/usr/bin/env python3
"""How much of the mesh tails is the view-INDEPENDENT neighbour walk (both Laplacians, edge hinge, delta term)?  The metric scene's
step with the regularisers evaluated for all vertices and for none (reg_range = (0, 0): the walk of k_mesh_tail_a and the L^T r walk's
inputs are zero rows).  HIP events per launch group.      python3 tools/micro/tail_reg_share.py [--views 48]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from fmhr_amd import synthetic                  # noqa: E402
from fmhr_amd.ham import HamStage2, VertexOrder, edge_length_mean, render_forward   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    scene = synthetic.scene_from_config("metric", 0, args.views)
    order = VertexOrder(scene["faces"], scene["V"])
    for k in ("vertices", "vertices_true", "albedo_true", "albedo0"):
        scene[k] = order.to_internal(scene[k])
    scene["faces"] = order.faces
    n, H, W, V = scene["views"], scene["H"], scene["W"], scene["V"]
    faces = scene["faces"].to(dev)
    w2c, proj, sh = scene["w2c"].to(dev).contiguous(), scene["proj"].to(dev).contiguous(), scene["sh0"].to(dev).contiguous()
    vtmp = scene["vertices"].to(dev).contiguous()
    tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    hs = HamStage2(faces, V, n, H, W, scene["weights"], edge_length_mean(vtmp, faces), n_global=n)
    delta = 1e-4 * torch.randn(V, 3, device=dev)
    albedo = scene["albedo0"].to(dev).contiguous()
    stream = torch.cuda.current_stream()
    slot = bench.SLOTS.index("regularisers")
    for name, rr in (("all vertices", None), ("no vertex", (0, 0)), ("all vertices", None)):
        for _ in range(5):
            hs.step(vtmp, delta, albedo, sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], reg_range=rr)
        K = 20
        tk = bench.Timing(K, stream)
        for i in range(K):
            tk.attach(i)
            hs.step(vtmp, delta, albedo, sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], reg_range=rr)
            bench.Timing.detach()
        torch.cuda.synchronize()
        print("%-13s  mesh tails %.1f us (event pair ~4.7 us included)" % (name, sum(tk.ms(i, slot) for i in range(K)) / K * 1e3))


if __name__ == "__main__":
    main()
