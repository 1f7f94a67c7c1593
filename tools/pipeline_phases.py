"""Where the wall time of an end-to-end HAM run goes (48 views x 512^2, sub3-class mesh, 150 stage-1 + 300 stage-2 iterations): the phases of
fmhr_amd.pipeline.run_ham timed one by one (rep 0 carries the one-off library / allocator warm-up).  python tools/pipeline_phases.py"""
import sys, time, torch
sys.path.insert(0, '.')
from fmhr_amd import synthetic, pipeline, stage1, optim
from fmhr_amd.ham import render_forward, HamStage2, edge_length_mean
views = 48
sc = synthetic.scene_from_config("metric", 0, views)
dev = torch.device("cuda:0")
H, W = sc["H"], sc["W"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
img = tg["img"]; gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
vt = sc["vertices"].to(dev)
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"{name:28s} {1e3*(time.perf_counter()-t):8.2f} ms"); return r
for rep in range(2):
    print("--- rep", rep)
    hs1 = T("HamStage1 (constant part+plan)", lambda: stage1.HamStage1(vt, faces, w2c, proj, img, tg["mask"], (H, W), 50.0, want_pred_masks=True))
    init = T("init_sh", lambda: stage1.init_sh(hs1, gray))
    elm = T("edge_length_mean", lambda: edge_length_mean(vt, faces))
    hs2 = T("HamStage2 construction", lambda: HamStage2(faces, vt.shape[0], views, H, W, sc["weights"], elm))
    sh = init["sh_coeffs"].clone().requires_grad_(True)
    alb = (torch.zeros_like(vt) + init["albedo_mean"]).unsqueeze(0).contiguous().requires_grad_(True)
    opt = optim.Adam([{'params': alb, 'lr': 5e-3}, {'params': sh, 'lr': 5e-3}])
    T("150 stage-1 steps", lambda: [hs1.step(alb, sh, optimizer=opt, params=(alb, sh)) for _ in range(150)])
    delta = torch.zeros_like(vt).requires_grad_(True)
    opt2 = optim.Adam([{'params': delta, 'lr': 1e-4}, {'params': alb, 'lr': 5e-3}])
    shf = sh.detach()
    T("300 stage-2 steps", lambda: [hs2.step(vt, delta.detach(), alb.detach()[0], shf, w2c, proj, img, tg["mask"], init["valid_masks"], optimizer=opt2, params=(delta, alb)) for _ in range(300)])
    T("300 x randperm().to(dev)", lambda: [torch.randperm(views).to(dev) for _ in range(300)])
