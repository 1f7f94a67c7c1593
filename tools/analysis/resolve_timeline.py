"""Wave timeline of the resolve pass (measurement build -DFMHR_X_TS: every wave leaves 100-MHz timestamps of its dependent
steps).  Build: FMHR_FORCE_BUILD=1 FMHR_EXTRA_FLAGS=-DFMHR_X_TS fmhr_amd/csrc/build.sh; run on the GPU box:
    python tools/analysis/resolve_timeline.py > gpurun_out/resolve_timeline.txt
"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from fmhr_amd import synthetic
from fmhr_amd.ham import HamStage2, edge_length_mean, render_forward
from fmhr_amd.optim import Adam

dev = torch.device("cuda:0")
sc = synthetic.scene_from_config("metric", 0)
faces = sc["faces"].to(dev)
w2c, proj, sh = sc["w2c"].to(dev).contiguous(), sc["proj"].to(dev).contiguous(), sc["sh0"].to(dev).contiguous()
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, 512, 512), device=dev)
vt = sc["vertices"].to(dev).contiguous()
V = sc["V"]
hs = HamStage2(faces, V, 48, 512, 512, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
for it in range(40):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
torch.cuda.synchronize()
N = 196608                                    # waves of the metric step: 48 * 512 * 512 / 64
buf = np.zeros(N * 8, dtype=np.uint64)
hs.lib.fmhr_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = hs.lib.fmhr_debug_ts(buf.ctypes.data, buf.nbytes)
assert rc == 0
t = buf.reshape(N, 8).astype(np.int64)
np.save("gpurun_out/resolve_ts.npy", t)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0                # 100 MHz -> microseconds
start, end = us(t[:, 0]), us(t[:, 5])
dirty = t[:, 2] != 0
print(f"sampled waves {N}, dirty {dirty.sum()} ({100 * dirty.mean():.1f} %), kernel span {end.max():.1f} us (first start 0, last end)")
life = end - start
print(f"clean waves: life mean {life[~dirty].mean():.2f} us, median {np.median(life[~dirty]):.2f}, p90 {np.percentile(life[~dirty], 90):.2f};"
      f" dirty-flag wait (t1 - t0) mean {(us(t[~dirty, 1]) - start[~dirty]).mean():.2f}")
d = t[dirty]
ds = us(d[:, 0])
hop = lambda a, b: (d[:, b] - d[:, a]) / 100.0
covered = d[:, 3] != 0                         # at least one lane had a key (always, for a dirty segment)
print(f"dirty waves: life mean {life[dirty].mean():.2f} us, median {np.median(life[dirty]):.2f}, p90 {np.percentile(life[dirty], 90):.2f}, max {life[dirty].max():.2f}")
for name, a, b in (("dirty flag", 0, 1), ("keys + mask", 1, 2), ("tri", 2, 3), ("verts + attrs", 3, 4), ("compute + stores issued", 4, 5), ("stores acknowledged", 5, 7)):
    h = hop(a, b)[covered]
    print(f"  {name:26s} mean {h.mean():.2f} us  median {np.median(h):.2f}  p90 {np.percentile(h, 90):.2f}")
print(f"sum of wave lifetimes: {life.sum() / 1e3:.1f} ms.wave -> mean resident {life.sum() / end.max():.0f} of 7168 slots")
# residency and launch progress over time
T = end.max()
edges = np.linspace(0, T, 21)
print("time slice (us)   waves started   resident at slice start   dirty share of started")
for i in range(20):
    a, b = edges[i], edges[i + 1]
    st = (start >= a) & (start < b)
    res = ((start <= a) & (end > a)).sum()
    print(f"  {a:6.1f}-{b:6.1f}   {st.sum():8d}   {res:8d}   {100 * dirty[st].mean() if st.any() else 0:6.1f} %")
# start time against wave index: is the dispatch in order and steady?
idx = np.arange(N)
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    k = int(q * N)
    print(f"wave {k:7d}: start {start[k]:.1f} us")
hw = t[:, 6] & 0xffffffff
xcc = (t[:, 6] >> 32) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
key = xcc * 256 + cu
uniq, cnt = np.unique(key, return_counts=True)
busy = np.array([life[key == k].sum() for k in uniq])
print(f"CUs seen {len(uniq)}; sampled waves per CU min/mean/max {cnt.min()}/{cnt.mean():.0f}/{cnt.max()}; wave-time per CU (us) min/mean/max {busy.min():.0f}/{busy.mean():.0f}/{busy.max():.0f}")
last_end = np.array([end[key == k].max() for k in uniq])
print(f"last wave end per CU: min {last_end.min():.1f} mean {last_end.mean():.1f} max {last_end.max():.1f} us")

# physical wave slots: how long does a slot sit empty between two waves?
simd, wslot = (hw >> 4) & 3, hw & 0xf
slot = key * 64 + simd * 16 + wslot
order = np.lexsort((start, slot))
s_slot, s_start, s_end = slot[order], start[order], end[order]
same = s_slot[1:] == s_slot[:-1]
gap = (s_start[1:] - s_end[:-1])[same]
print(f"physical slots used {len(np.unique(slot))}; gap between a wave's end and the next wave's start on the same slot: mean {gap.mean():.2f} us, median {np.median(gap):.2f}, p90 {np.percentile(gap, 90):.2f}")
busy_frac = life.sum() / (len(np.unique(slot)) * end.max())
print(f"slot busy fraction {busy_frac:.2f}")
# the four waves of a workgroup: spread of their start times
st4 = start.reshape(-1, 4)
print(f"start spread within a workgroup: mean {(st4.max(1) - st4.min(1)).mean():.2f} us")
# workgroup dispatch gaps per XCD (workgroups of one XCD in index order)
wg_start = st4.min(1)
for x in range(2):
    g = np.diff(wg_start[x::8])
    print(f"XCD {x}: gap between consecutive workgroup starts mean {g.mean()*1000:.0f} ns, median {np.median(g)*1000:.0f} ns, p90 {np.percentile(g, 90)*1000:.0f} ns, negative {100*(g<0).mean():.0f} %")
