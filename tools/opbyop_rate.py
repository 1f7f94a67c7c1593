"""The import-swap experience: the reference's stage-2 loop body mesh_sfs_optim.py:253-308, unchanged, on the generic
fmhr_amd.dr / fmhr_amd.utils ops with autograd — time per iteration on the metric workload, and which kernels it spends it in.
    python tools/opbyop_rate.py [views]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fmhr_amd import synthetic
from fmhr_amd.ham import render_forward, edge_length_mean
from fmhr_amd.examples.loop_body import loop_body_on_dr as _loop_body_on_dr
views = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sc = synthetic.scene_from_config("metric", 0, views)
dev = torch.device("cuda:0")
H, W = sc["H"], sc["W"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
g = dict(vertices=sc["vertices"].to(dev), delta=torch.zeros_like(sc["vertices"]).to(dev), albedo=sc["albedo0"].to(dev), sh0=sc["sh0"].to(dev),
         w2c=w2c, proj=proj, img=tg["img"], mask=tg["mask"], valid_mask=tg["valid_mask"])
elm = edge_length_mean(g["vertices"], faces)
for _ in range(2): _loop_body_on_dr(g, faces, views, H, W, sc["weights"], elm)
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K): _loop_body_on_dr(g, faces, views, H, W, sc["weights"], elm)
torch.cuda.synchronize()
print(f"op-by-op stage-2 iteration (loop body unchanged, generic ops + autograd), {views} views {H}x{W}: {1e3 * (time.perf_counter() - t0) / K:.1f} ms")
