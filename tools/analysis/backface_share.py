"""How much of the rasterizer's work goes into BACK-FACING fragments (round 6, for the 'what comes next' list)?  Both orientations are
drawn (rule R3: no culling), but on a closed surface the nearest fragment of a pixel is front-facing; the hand mesh is open only at the
wrist.  After a pre-roll of the metric workload this reads the step's coverage entries (candidate masks per (view, face)), owner marks and
snapped vertices from the workspace and counts fragments and winning (view, face)s by screen-space orientation.
    gpurun -- 'python tools/analysis/backface_share.py [config] [steps]'"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, VertexOrder, edge_length_mean, render_forward
from fmhr_amd.optim import Adam

cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
sc = synthetic.scene_from_config(cfg, 0)
order = VertexOrder(sc["faces"], sc["V"])
for k in ("vertices", "vertices_true", "albedo_true", "albedo0"):
    sc[k] = order.to_internal(sc[k])
sc["faces"] = order.faces
n, H, W, V, F = sc["views"], sc["H"], sc["W"], sc["V"], sc["F"]
faces = sc["faces"].to(dev)
w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
vt = sc["vertices"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
hs = HamStage2(faces, V, n, H, W, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
for _ in range(steps):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
torch.cuda.synchronize()
hs.check_overflow()
cov = hs.raw("cov16", torch.int32, n * F * 2).view(n, F, 2)
own = hs.raw("own8", torch.uint8, n * F).view(n, F)
snap = hs.raw("snap", torch.int32, n * V * 2).view(n, V, 2).long()
stamp = int(hs.raw("stamp", torch.int32, 4)[0])
fl = faces.long()
a, b, c = snap[:, fl[:, 0]], snap[:, fl[:, 1]], snap[:, fl[:, 2]]
area2 = (b[..., 0] - a[..., 0]) * (c[..., 1] - a[..., 1]) - (b[..., 1] - a[..., 1]) * (c[..., 0] - a[..., 0])
# this step's entries: the stamp rides in the second word of LARGE-box markers only; a small-box entry is this step's if its owner mark is
# (owners) — for the fragment count every (view, face) is re-evaluated from its mask as written in this step (entries are rewritten by every
# covering triangle in every step; a triangle that stopped covering keeps a stale entry, which the candidate count below would over-count:
# restrict to faces whose snapped box is on screen and whose area is non-zero, as the kernel does)
width = (cov[..., 1].long() >> 26) & 63
small = width != 63
frags = torch.zeros(n, F, dtype=torch.long, device=dev)
m = cov[..., 0].long() & 0xFFFFFFFF
for bit in range(32):
    frags += (m >> bit) & 1
frags = torch.where(small & (area2 != 0), frags, torch.zeros_like(frags))
front = area2 > 0          # (screen y grows downwards: which sign is "front" depends on the convention — report both, the larger owner share names it)
owner = own == (stamp & 0xFF)
res = {}
for name, sel in (("area2>0", front), ("area2<0", area2 < 0)):
    res[name] = (int(frags[sel].sum()), int((owner & sel).sum()))
tot_f, tot_o = sum(v[0] for v in res.values()), sum(v[1] for v in res.values())
print(f"{cfg}: {n} views, F = {F}, after {steps} steps: fragments (candidate pixel centres of small-box triangles) {tot_f}, owner (view, face)s {tot_o}")
for name, (f_, o_) in res.items():
    print(f"  {name}: fragments {f_} ({100.0 * f_ / max(tot_f, 1):.1f} %), owners {o_} ({100.0 * o_ / max(tot_o, 1):.1f} %)")
