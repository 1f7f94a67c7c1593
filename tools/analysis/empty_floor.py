"""Floors of the step's kernels: the metric workload with every view looking past the mesh (clip x shifted by 5 w: every triangle is
off screen, no fragment, no owner) — what the launches cost before any pixel work.  Run under rocprofv3 --kernel-trace --stats.
    python tools/analysis/empty_floor.py [shift]      (shift 0 = the normal scene, for the same script's own reference)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, render_forward, edge_length_mean
shift = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
sc = synthetic.scene_from_config("metric", 0, None)
dev = torch.device("cuda:0")
H, W, V, n = sc["H"], sc["W"], sc["V"], sc["views"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
proj = proj.clone(); proj[:, :, 0] += shift * proj[:, :, 3]
vt = sc["vertices"].to(dev).contiguous()
hs = HamStage2(faces, V, n, H, W, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev); albedo = sc["albedo0"].to(dev).contiguous(); sh = sc["sh0"].contiguous().to(dev)
for _ in range(30):
    hs.step(vt, delta, albedo, sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
torch.cuda.synchronize()
print("done", hs.loss_dict())
