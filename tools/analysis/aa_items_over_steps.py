import torch, sys, ctypes
sys.path.insert(0, ".")
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
dev = torch.device("cuda:0")
sc = synthetic.scene_from_config("metric", 0)
faces = sc["faces"].to(dev); w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, 512, 512), device=dev)
vt = sc["vertices"].to(dev).contiguous()
V = sc["V"]
hs = HamStage2(faces, V, 48, 512, 512, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
o = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"aa_counter")
for it in range(241):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
    if it in (0, 10, 30, 60, 120, 240):
        torch.cuda.synchronize()
        rc = hs.ws[o + 256: o + 256 + 768 * 4].view(torch.int32)          # 768 regions when W % 64 == 0 (antialias.hip: aa_grid)
        print("step", it, "aa items", int(rc.sum()), "max per region", int(rc.max()), "delta max", float(delta.abs().max()))
