"""How well are k_face_bwd's lanes used?  Owned pixels per (view, face) from the CPU checker's rasterizer on the metric scene
(analysis only: the checker is test infrastructure), then wave-iteration counts of the task loop for three assignments:
  current  : 8 lanes per face, a wave = 8 consecutive faces, iterations = max over its faces of ceil(T / 8)
  sorted   : the 32 faces of a workgroup sorted by T before they are dealt to the 4 waves
  pooled   : a wave's 64 lanes share its 8 faces' tasks: ceil(sum T / 64)
    python tools/analysis/face_bwd_balance.py [views]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import mesh_scene
from oracle import ham_oracle as O
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = mesh_scene("metric", 0, nv)
F, H, W = sc["F"], sc["H"], sc["W"]
rast, _ = O.rasterize(O.RasterizeGLContext() if hasattr(O, "RasterizeGLContext") else None, sc["pos"][:nv].float(), sc["faces"], (H, W))
ids = rast[..., 3].long().reshape(nv, -1)
T = np.zeros(F, dtype=np.int64)
owners = 0
for b in range(nv):
    c = torch.bincount(ids[b], minlength=F + 1)[1:].numpy()
    T += c
    owners += int((c > 0).sum())
print(f"{nv} views: owned pixels {T.sum()}, owning (view, face) pairs {owners} of {nv * F} ({owners / (nv * F):.3f}), per face mean {T.mean():.1f} max {T.max()}")
n32 = F // 32 * 32
Tb = T[:n32].reshape(-1, 32)
it = lambda x: -(-x // 8)
cur = it(Tb.reshape(-1, 4, 8)).max(2).sum()
srt = it(np.sort(Tb, 1).reshape(-1, 4, 8)).max(2).sum()
pool = (-(-Tb.reshape(-1, 4, 8).sum(2) // 64)).sum()
ideal = -(-T.sum() // 64)
print(f"wave-iterations of the pixel loop: current {cur}, sorted within the workgroup {srt} ({srt / cur:.2f}x), pooled per wave {pool} ({pool / cur:.2f}x), ideal {ideal} ({ideal / cur:.2f}x)")
print("lane utilisation now: %.2f" % (T[:n32].sum() / (cur * 64)))
# 4 lanes per face (a wave = 16 consecutive faces): fewer idle lanes per face, but the wave waits for the busiest of 16
n16 = F // 16 * 16
it4 = (-(-T[:n16].reshape(-1, 16) // 4)).max(1).sum()
print(f"4 lanes per face: {it4} wave-iterations over {n16 // 16} waves ({it4 / cur:.2f}x the lane-time of the current form)")
it16 = (-(-T[:n32].reshape(-1, 4) // 16)).max(1).sum()
print(f"16 lanes per face: {it16} wave-iterations ({it16 / cur:.2f}x)")
