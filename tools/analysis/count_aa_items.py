import torch, sys, ctypes
sys.path.insert(0, ".")
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
dev = torch.device("cuda:0")
sc = synthetic.scene_from_config("metric", 0)
faces = sc["faces"].to(dev); w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, 512, 512), device=dev)
vt = sc["vertices"].to(dev).contiguous()
hs = HamStage2(faces, sc["V"], 48, 512, 512, sc["weights"], edge_length_mean(vt, faces))
hs.step(vt, torch.zeros_like(vt), sc["albedo0"].to(dev).contiguous(), sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"])
torch.cuda.synchronize()
o = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"aa_counter")
rc = hs.ws[o + 256: o + 256 + 8192].view(torch.int32)
print("aa items:", int(rc.sum()), "regions used:", int((rc > 0).sum()), "max per region:", int(rc.max()))
oi = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"aa_items")
cap = hs.dims.aa_cap; region = cap // 1280
items = hs.ws[oi: oi + cap * 16].view(torch.int32).view(cap, 4)
va = []
for r in range(1280):
    c = int(rc[r])
    if c: va.append(items[r * region: r * region + c])
it = torch.cat(va).cpu()
import numpy as np
v = np.concatenate([it[:, 2].numpy(), it[:, 3].numpy()])
u, cnt = np.unique(v, return_counts=True)
print("distinct vertices:", len(u), "refs:", len(v), "mean refs/vertex:", cnt.mean(), "max:", cnt.max())
# per-region distinct vertices
d = []
for x in va[:200]:
    x = x.cpu().numpy(); vv = np.concatenate([x[:, 2], x[:, 3]]); d.append(len(np.unique(vv)) / len(vv))
print("per-region distinct/refs:", np.mean(d))
v = hs.views()
pm = v["pred_mask"]
print("pixels with fractional coverage:", int(((pm > 0) & (pm < 1)).sum()), "covered:", int((v["rast"][..., 3] > 0).sum()))
print("aa_counter:", hs.ws[o: o + 8].view(torch.int32).tolist(), "codes clamped:", int(((it[:, 1] >> 16) & 32 != 0).sum()))
print("rc[:16]", rc[:16].tolist(), "cap", cap, "region", region, "sum first 1280:", int(rc[:1280].sum()), "sum all 2048:", int(rc.sum()))
o2 = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"g_shaded")
print("offsets aa_counter", o, "g_shaded", o2, "diff", o2 - o)
