"""Per-step device time of consecutive steps (HIP events around every step) of one configuration after bench.py's own pre-roll
logic is bypassed: 200 untimed steps, then N timed ones — looks for periodic slow steps.   python tools/analysis/step_times.py C4 80"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, render_forward, edge_length_mean
from fmhr_amd.optim import Adam
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
sc = synthetic.scene_from_config(cfg, 0, None)
dev = torch.device("cuda:0")
H, W, V, n = sc["H"], sc["W"], sc["V"], sc["views"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
vt = sc["vertices"].to(dev).contiguous()
hs = HamStage2(faces, V, n, H, W, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True); albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
sh = sc["sh0"].contiguous().to(dev)
lr = sc["lr"]
opt = Adam([{'params': delta, 'lr': lr["lr"]}, {'params': albedo, 'lr': lr["albedo_lr"]}])
def step():
    return hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
for _ in range(200): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
s = torch.cuda.current_stream()
ev[0].record(s)
for i in range(N):
    step(); ev[i + 1].record(s)
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print(cfg, "median %.3f ms, max %.3f ms at step %d" % (sorted(t)[N // 2], max(t), t.index(max(t))))
print(" ".join("%.2f" % x for x in t))
