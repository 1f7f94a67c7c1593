"""debug: where does the fused step's albedo gradient differ from the fp32 oracle twin on the demo-capture scene?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ham_oracle as oracle
from test_gpu_ham import demo_capture_scene, demo_capture_targets
from fmhr_amd.ham import HamStage2
dev = torch.device("cuda:0")
sc, g, cams, X = demo_capture_scene()
sc = demo_capture_targets(sc, dev)
gk = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
def run(**kw):
    hs = HamStage2(sc["faces"].to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], sc["elm"], **kw)
    hs.step(gk["vertices"], gk["delta"], gk["albedo"], gk["sh0"], gk["w2c"], gk["proj"], gk["img"], gk["mask"], gk["valid_mask"])
    v = hs.views()
    return hs, hs.gdelta.clone().cpu(), hs.galbedo.clone().cpu(), v["rast"].clone().cpu(), v["pos"].clone().cpu()
hs, gd, ga, rast, pos = run()
print("binned", hs.binned, "pitch", hs.pitch)
d32 = sc["delta"].clone().requires_grad_(True)
a32 = sc["albedo"][None].clone().requires_grad_(True)
ref = oracle.ham_stage2_losses(sc["vertices"], d32, a32, sc["sh0"], sc["w2c"], sc["proj"], sc["img"], sc["mask"], sc["valid_mask"], sc["faces"],
                               (sc["H"], sc["W"]), sc["weights"], sc["elm"], pos_value=pos)
ref["loss"].backward()
ga32 = a32.grad[0]
print("rast equal", torch.equal(rast, ref["rast"]))
err = (ga - ga32).abs().max(1).values
print("max |galbedo| ref", float(ga32.abs().max()), "max err", float(err.max()), "n(err > 1e-3 max)", int((err > 1e-3 * ga32.abs().max()).sum()))
bad = torch.argsort(err, descending=True)[:12]
ids = rast[..., 3].long()
faces = sc["faces"].long()
for v in bad.tolist():
    fs = torch.nonzero((faces == v).any(1)).flatten()
    info = []
    for f in fs.tolist():
        for b in range(2):
            ys, xs = torch.nonzero(ids[b] == f + 1, as_tuple=True)
            if len(ys):
                info.append((f, b, len(ys), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)))
    print("vertex", v, "gpu", ga[v].tolist(), "ref", ga32[v].tolist(), "faces (f, view, owned px, bw, bh):", info)
# which form reproduces it?
os.environ["FMHR_RASTER_BINNED"] = "0"
hs2, gd2, ga2, rast2, _ = run()
print("atomic form: max err vs ref", float((ga2 - ga32).abs().max()), "vs binned", float((ga2 - ga).abs().max()))
del os.environ["FMHR_RASTER_BINNED"]
# box statistics of owners
for b in range(2):
    idb = ids[b]
    ys, xs = torch.nonzero(idb > 0, as_tuple=True)
    fid = idb[ys, xs]
    F = sc["F"]
    mnx = torch.full((F + 1,), 10**6).scatter_reduce(0, fid, xs, "amin"); mxx = torch.full((F + 1,), -1).scatter_reduce(0, fid, xs, "amax")
    mny = torch.full((F + 1,), 10**6).scatter_reduce(0, fid, ys, "amin"); mxy = torch.full((F + 1,), -1).scatter_reduce(0, fid, ys, "amax")
    own = mxx >= 0
    box = ((mxx - mnx + 1) * (mxy - mny + 1))[own]
    cnt = torch.bincount(fid, minlength=F + 1)[own]
    print("view", b, "owners", int(own.sum()), "owned px/face mean", float(cnt.float().mean()), "max", int(cnt.max()), "owned-box > 32:", int((box > 32).sum()), "max box", int(box.max()))
