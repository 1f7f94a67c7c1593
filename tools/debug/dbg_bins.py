import os, sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import ham_oracle as oracle
from test_gpu_ham import prepare
from fmhr_amd.ham import HamStage2
oracle.lib()
sc = prepare(oracle, (4, 192, 192, 64, 96), 2)
ring = lambda r, s: 1 + r * 96 + s
big = torch.tensor([[ring(63, 0), ring(63, 32), ring(63, 64)], [ring(63, 0), ring(63, 64), ring(63, 80)]], dtype=torch.int32)
f = torch.cat([sc["faces"], big]).contiguous()
sc = dict(sc, faces=f, F=f.shape[0])
dev = torch.device("cuda:0")
g = {k: sc[k].to(dev).contiguous() for k in ("vertices", "delta", "albedo", "sh0", "w2c", "proj", "img", "mask", "valid_mask")}
args = (g["vertices"], g["delta"], g["albedo"], g["sh0"], g["w2c"], g["proj"], g["img"], g["mask"], g["valid_mask"])
out = {}
for mode in ("1", "0"):
    os.environ["FMHR_RASTER_BINNED"] = mode
    hs = HamStage2(f.to(dev), sc["V"], sc["views"], sc["H"], sc["W"], sc["weights"], 1e-6)
    for it in range(2):
        hs.step(*args)
        v = hs.views()
        out[(mode, it)] = v["rast"].clone()
        out[(mode, it, "img")] = v["tmp_img"].clone()
        out[(mode, it, "pm")] = v["pred_mask"].clone()
        out[(mode, it, "pre")] = v["tmp_img_pre_aa"].clone()
        out[(mode, it, "w")] = hs.raw("pix16", torch.int32, 4 * 192 * 192 * 4).view(4, 192, 192, 4)[..., 3].clone()
        out[(mode, it, "aa")] = int(hs.aa_item_count())
    print(mode, hs.binned, hs.raster_overflowed(), int(hs.raw("bin_cursor", torch.int32, 16 * 4 * 192 * 3)[::16].max()))
for it in range(2):
    a, b = out[("1", it)], out[("0", it)]
    d = (a != b).any(-1)
    print("step", it, "differing pixels", int(d.sum()))
    idx = torch.nonzero(d)[:8]
    for i in idx:
        print(i.tolist(), a[tuple(i)].tolist(), b[tuple(i)].tolist())
print("binned step0 vs step1 differ:", int((out[("1", 0)] != out[("1", 1)]).any(-1).sum()))

for it in range(2):
    for k in ("img", "pm", "pre"):
        a, b = out[("1", it, k)], out[("0", it, k)]
        d = (a != b)
        d = d.any(-1) if d.dim() == 4 else d
        print("step", it, k, "differing pixels", int(d.sum()), "max abs", float((a - b).abs().max()))
    cov = out[("1", it)][..., 3] > 0
    wa, wb = out[("1", it, "w")], out[("0", it, "w")]
    dw = (wa != wb) & cov
    print("step", it, "packed word differs on covered px:", int(dw.sum()), "aa items", out[("1", it, "aa")], out[("0", it, "aa")])
    for i in torch.nonzero(dw)[:6]:
        print("   ", i.tolist(), hex(int(wa[tuple(i)]) & 0xffffffff), hex(int(wb[tuple(i)]) & 0xffffffff))
