"""Time one stage-1 ("albedo") iteration of fmhr_amd.pipeline (mesh_sfs_optim.py:205-236) on the metric-size mesh."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from fmhr_amd import synthetic, pipeline
from fmhr_amd.ham import render_forward
from fmhr_amd.optim import Adam
views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = synthetic.scene_from_config("metric", 0, views)
dev = torch.device("cuda:0")
n, H, W, V = sc["views"], sc["H"], sc["W"], sc["V"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev); sh = sc["sh0"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
vt = sc["vertices"].to(dev).contiguous()
ops = pipeline.product_ops()
glctx = ops.RasterizeGLContext()
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
shc = sh.clone().requires_grad_(True)
opt = Adam([{'params': albedo, 'lr': 5e-3}, {'params': shc, 'lr': 5e-3}])
cache = pipeline.Stage1Cache(ops, glctx, vt, faces, w2c, proj, tg["mask"], (H, W)) if (len(sys.argv) < 3 or sys.argv[2] not in ("nocache", "fusedonly")) else None
fusedonly = len(sys.argv) > 2 and sys.argv[2] == "fusedonly"
def step():
    if cache is not None:
        loss = pipeline.stage1_loss_cached(ops, cache, torch.arange(n, device=dev), albedo, shc, tg["img"], 50.0)
    else:
        loss = pipeline.stage1_loss(ops, glctx, vt, faces, albedo, shc, w2c, proj, tg["img"], tg["mask"], (H, W), 50.0)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
K = 30 if not fusedonly else 0
for _ in range(5 if not fusedonly else 0): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): l = step()
torch.cuda.synchronize()
if not fusedonly: print(f"stage-1 step, {n} views {H}x{W}, V={V}: {1e3*(time.perf_counter()-t0)/K:.3f} ms/step (cached rasterization: {cache is not None}), loss {float(l):.5f}")

# the fused form (fmhr_amd.stage1.HamStage1: plan built once, two launches per iteration, Adam inside)
from fmhr_amd.stage1 import HamStage1
t0 = time.perf_counter()
hs = HamStage1(vt, faces, w2c, proj, tg["img"], tg["mask"], (H, W), 50.0)
torch.cuda.synchronize()
t_plan = time.perf_counter() - t0
albedo2 = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
sh2 = sh.clone().requires_grad_(True)
opt2 = Adam([{'params': albedo2, 'lr': 5e-3}, {'params': sh2, 'lr': 5e-3}])
for _ in range(5): hs.step(albedo2, sh2, optimizer=opt2, params=(albedo2, sh2))
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K): l2, _, _ = hs.step(albedo2, sh2, optimizer=opt2, params=(albedo2, sh2))
torch.cuda.synchronize()
print(f"fused stage-1 step: {1e3*(time.perf_counter()-t0)/K:.4f} ms/step (plan: {t_plan*1e3:.0f} ms once; rows {hs.rows}, entries {hs.entries}, antialias items {hs.n_items}), loss {float(l2.sum()):.5f}")
