import time, torch, sys
sys.path.insert(0, '/root/repo')
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
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
        l, gd, ga = hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
        delta.grad, albedo.grad = gd, ga.view(1, V, 3)
        opt.step()
    for _ in range(20): step()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"views {views}: host enqueue {1e3*(t1-t0)/K:.4f} ms/step, total {1e3*(t2-t0)/K:.4f} ms/step")
