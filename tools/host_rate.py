"""tools/host_rate.py — how long does the HOST take to enqueue one step?  (plain step; with FMHR_FORCE_SHARDED=1 the view-sharded step
with its all-reduce over RCCL, world of one: through torch.distributed, or FMHR_STREAM_ALLREDUCE=1 straight into RCCL)"""
import os, time, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
group = None
if os.environ.get("FMHR_FORCE_SHARDED") == "1":
    from fmhr_amd import dist as fdist
    fdist.init_from_env()
    torch.cuda.set_device(0)
    group = fdist.make_stream_all_reduce(torch.distributed.group.WORLD)
    print("group:", type(group).__name__)
for views in (6, 48):
    sc = synthetic.scene_from_config("metric", 0, views)
    dev = torch.device("cuda:0")
    n, H, W, V = sc["views"], sc["H"], sc["W"], sc["V"]
    faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev); sh = sc["sh0"].contiguous().to(dev)
    tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
    vt = sc["vertices"].to(dev).contiguous()
    hs = HamStage2(faces, V, n, H, W, sc["weights"], edge_length_mean(vt, faces))
    delta = torch.zeros(V, 3, device=dev, requires_grad=True)
    albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
    opt = Adam([{'params': delta, 'lr': 1e-4}, {'params': albedo, 'lr': 1e-3}])
    def step():
        if group is None and os.environ.get("HOST_RATE_SEPARATE_ADAM") == "1":
            l, gd, ga = hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
            delta.grad, albedo.grad = gd, ga.view(1, V, 3)
            opt.step()
        else:
            hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], group=group,
                    reg_range=(0, V), optimizer=opt, params=(delta, albedo))
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"views {views}: host enqueue {1e3*(t1-t0)/K:.4f} ms/step, total {1e3*(t2-t0)/K:.4f} ms/step")
