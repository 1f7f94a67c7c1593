import torch, sys, ctypes
sys.path.insert(0, ".")
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam
dev = torch.device("cuda:0")
sc = synthetic.scene_from_config("metric", 0)
faces = sc["faces"].to(dev); w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, 512, 512), device=dev)
vt = sc["vertices"].to(dev).contiguous(); V = sc["V"]
hs = HamStage2(faces, V, 48, 512, 512, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
oa = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"act"); 
for it in range(100):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
torch.cuda.synchronize()
# act_list / act_count offsets are not exported by name: read through the layout order (act, vmnz, act_list, act_count)
nseg = 48 * 512 * 512 // 64
al = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"act")
rnd = lambda x: (x + 255) & ~255
off_list = al + 2 * rnd(nseg)
lst = hs.ws[off_list: off_list + nseg * 4].view(torch.int32)
cnt = hs.ws[off_list + rnd(nseg * 4): off_list + rnd(nseg * 4) + 8].view(torch.int32)
c = int(cnt[0]); l = lst[:c]
print("active segments", c, "of", nseg, "; to scan (flag set)", int((l < 0).sum()), "; skipped", int((l >= 0).sum()))
# silhouette-pixel words (resolve pass): how many pixels per scanned segment can start a blending pair?
import numpy as np
osil = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"silw")
ocov = hs.lib.fmhr_ham_workspace_offset(ctypes.byref(hs.dims), b"covw")
if osil >= 0:
    sil = hs.ws[osil: osil + nseg * 8].view(torch.int64).cpu().numpy().view(np.uint64)
    cov = hs.ws[ocov: ocov + nseg * 8].view(torch.int64).cpu().numpy().view(np.uint64)
    segs = (l[l < 0] & 0x7fffffff).cpu().numpy()
    pc = lambda a: np.array([bin(int(x)).count("1") for x in a])
    spr = 512 // 64
    s0 = sil[segs]; sR = np.where((segs % spr) == spr - 1, 0, sil[np.minimum(segs + 1, nseg - 1)]); sD = sil[np.minimum(segs + spr, nseg - 1)]
    M = s0 | (s0 >> np.uint64(1)) | (sR << np.uint64(63)) | sD
    print("per scanned segment: covered pixels", pc(cov[segs]).mean(), "silhouette-face pixels", pc(s0).mean(), "candidate pixels (mask M)", pc(M).mean(),
          "; segments with M == 0:", int((M == 0).sum()), "of", len(segs))
