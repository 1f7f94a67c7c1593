import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, VertexOrder, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
dev = torch.device("cuda:0")
scene = synthetic.scene_from_config("metric", 0, None)
order = VertexOrder(scene["faces"], scene["V"])
for k in ("vertices", "vertices_true", "albedo_true", "albedo0"): scene[k] = order.to_internal(scene[k])
scene["faces"] = order.faces
n, H, W, V = scene["views"], scene["H"], scene["W"], scene["V"]
faces = scene["faces"].to(dev)
w2c, proj, sh = scene["w2c"].to(dev).contiguous(), scene["proj"].to(dev).contiguous(), scene["sh0"].to(dev).contiguous()
vt = scene["vertices"].to(dev).contiguous()
tg = synthetic.make_targets(scene, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
hs = HamStage2(faces, V, n, H, W, scene["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True); albedo = scene["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': scene["lr"]["lr"]}, {'params': albedo, 'lr': scene["lr"]["albedo_lr"]}])
for _ in range(200):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
torch.cuda.synchronize()
nseg = n * H * hs.pitch // 64
act = hs.raw("act", torch.uint8, nseg).clone()
covw = hs.raw("covw", torch.int64, nseg).clone()
a = act != 0
print("segments", nseg, "active", int(a.sum()), "active without coverage", int((a & (covw == 0)).sum()), "covered", int((covw != 0).sum()))
cov_px = sum(bin(int(x) & (2**64-1)).count("1") for x in covw[covw != 0][:20000].tolist()) / 20000
print("mean covered pixels per covered segment (first 20000): %.1f" % cov_px)
vm = tg["valid_mask"].reshape(-1, 64)
print("segments with valid_mask != 0:", int((vm != 0).any(1).sum()), " mask>0 segments:", int((tg["mask"].reshape(-1,64) > 0).any(1).sum()))
