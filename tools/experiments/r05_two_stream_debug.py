"""diagnose the two-stream mismatch: per-step losses of both runs, where rast differs, how far the parameters are apart"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
from fmhr_amd import synthetic as S
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
from util import rel_err
dev = torch.device("cuda:0")
n, H, W = 4, 256, 334
sc = S.make_scene(n, H, W, "mano", 0, 2, seed=5, hands=2)
V = sc["V"]
faces = sc["faces"].to(dev)
w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
vt = sc["vertices"].to(dev).contiguous()
elm = edge_length_mean(vt, faces)
fwd = lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W)
targets = []
for frame in range(2):
    scf = dict(sc)
    if frame:
        scf["vertices_true"] = sc["vertices"] + 0.0005 * torch.randn(sc["vertices"].shape, generator=torch.Generator().manual_seed(1000 + frame))
    targets.append(S.make_targets(scf, fwd, device=dev))

class Problem:
    def __init__(self, frame):
        self.tg = targets[frame]
        self.hs = HamStage2(faces, V, n, H, W, sc["weights"], elm)
        self.delta = torch.zeros(V, 3, device=dev, requires_grad=True)
        self.albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
        self.opt = Adam([{'params': self.delta, 'lr': sc["lr"]["lr"]}, {'params': self.albedo, 'lr': sc["lr"]["albedo_lr"]}])
        self.hist = []
    def step(self):
        l = self.hs.step(vt, self.delta.detach(), self.albedo.detach()[0], sh, w2c, proj, self.tg["img"], self.tg["mask"],
                         self.tg["valid_mask"], optimizer=self.opt, params=(self.delta, self.albedo))[0]
        self.hist.append(l.clone())

mode = sys.argv[1] if len(sys.argv) > 1 else "streams"
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    seq = [Problem(0), Problem(1)]
    for _ in range(6):
        for p in seq:
            p.step()
    torch.cuda.synchronize()
    con = [Problem(0), Problem(1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for it in range(6):
        cur = torch.cuda.current_stream()
        if mode == "streams":
            for s in streams: s.wait_stream(cur)
            for i, p in enumerate(con):
                with torch.cuda.stream(streams[i]): p.step()
            for s in streams: cur.wait_stream(s)
        else:
            for p in con: p.step()
    torch.cuda.synchronize()
    for f in range(2):
        ra, rb = seq[f].hs.views()["rast"].clone(), con[f].hs.views()["rast"].clone()
        same = torch.equal(ra, rb)
        dd = rel_err(con[f].delta.detach(), seq[f].delta.detach())
        hl = [float((a[:7] - b[:7]).abs().max() / b[:7].abs().max()) for a, b in zip(con[f].hist, seq[f].hist)]
        nv = [(float(a[7]), float(b[7])) for a, b in zip(con[f].hist, seq[f].hist)]
        msg = f"trial {trial} frame {f}: rast same {same} delta rel {dd:.1e} losses rel per step {['%.0e' % x for x in hl]}"
        if not same:
            diff = (ra != rb).any(-1)
            msg += f" | {int(diff.sum())} pixels differ, per view {diff.sum((1, 2)).tolist()}, ids differ {int((ra[..., 3] != rb[..., 3]).sum())}, n_valid {nv}"
        print(msg, flush=True)
