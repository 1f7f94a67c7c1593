"""Round 5: the FIRST run_ham of a process (cold allocator, code objects not loaded yet), phase by phase: wall-clock marks around the
library calls, each mark behind a synchronisation.    gpurun -- python tools/experiments/r05_run_ham_cold.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import pipeline, synthetic, stage1, ham                              # noqa: E402
from fmhr_amd.ham import render_forward                                           # noqa: E402

dev = torch.device("cuda:0")
sc = synthetic.make_scene(16, 1024, 1280, "mano", 0, 3, seed=0, hands=2)
H, W = sc["H"], sc["W"]
faces = sc["faces"].to(dev)
w2c, proj = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
img = tg["img"]
gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
marks = []


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        marks.append((name, 1e3 * (time.perf_counter() - t0)))
        return r
    setattr(mod, name, g)


wrap(stage1, "HamStage1")
wrap(stage1, "init_sh")
wrap(ham, "VertexOrder")
wrap(ham, "edge_length_mean")
conf = dict(num=16, batch=8, degree=3, epoch_albedo=150, epoch_sfs=300, sfs_weight=50.0, lap_weight=2000.0, albedo_weight=1.0,
            mask_weight=1000.0, edge_weight=500000.0, delta_weight=50000.0, lr=1e-4, albedo_lr=5e-3, sh_lr=5e-3)
for rep in range(2):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipeline.run_ham(sc["vertices"].to(dev), faces, img, gray, tg["mask"], w2c, proj, (H, W), conf, log_every=0)
    torch.cuda.synchronize()
    print(f"run {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms;", ", ".join(f"{n} {t:.1f}" for n, t in marks), flush=True)
