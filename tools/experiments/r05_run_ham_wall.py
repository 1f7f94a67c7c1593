"""Round 5: wall-clock of the WHOLE run the reference's demo does (conf/demo_sfs.conf: 16 views, batch 8, 1280 x 1024, two hands,
150 epochs of stage 1 + 300 of stage 2), through fmhr_amd.pipeline.run_ham on the synthetic two-hand scene — and where it goes:
cProfile of the host side (cumulative, top entries) next to the synchronised total.
    gpurun -- python tools/experiments/r05_run_ham_wall.py [epoch_albedo epoch_sfs]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import pipeline, synthetic                                          # noqa: E402
from fmhr_amd.ham import render_forward                                           # noqa: E402


def main():
    ea, es = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (150, 300)
    dev = torch.device("cuda:0")
    sc = synthetic.make_scene(16, 1024, 1280, "mano", 0, 3, seed=0, hands=2)
    H, W = sc["H"], sc["W"]
    faces = sc["faces"].to(dev)
    w2c, proj = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous()
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    img = tg["img"]
    gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
    conf = dict(num=16, batch=8, degree=3, epoch_albedo=ea, epoch_sfs=es, sfs_weight=50.0, lap_weight=2000.0, albedo_weight=1.0,
                mask_weight=1000.0, edge_weight=500000.0, delta_weight=50000.0, lr=1e-4, albedo_lr=5e-3, sh_lr=5e-3)
    verts = sc["vertices"].to(dev)
    for rep in range(2):                                             # the second run has warm allocators / loaded code objects
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        out = pipeline.run_ham(verts, faces, img, gray, tg["mask"], w2c, proj, (H, W), conf, log_every=0)
        torch.cuda.synchronize()
        pr.disable()
        t1 = time.perf_counter()
        print(f"run {rep}: {ea} + {es} epochs of 2 steps: {1e3 * (t1 - t0):.1f} ms wall, fullest bin {out['bin_depth_max']}", flush=True)
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
