"""Start / end / hardware slot of every wave of k_raster_tris, k_face_bwd, k_aa_fwd_list and k_aa_bwd_items (measurement build -DFMHR_X_TS): slot utilisation, the
tail of the launch and the balance over XCDs / CUs.  On the GPU box:
    FMHR_FORCE_BUILD=1 FMHR_EXTRA_FLAGS=-DFMHR_X_TS fmhr_amd/csrc/build.sh   (here, before gpurun)
    python tools/analysis/wave_timeline.py > gpurun_out/wave_timeline.txt
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
V, F = sc["V"], faces.shape[0]
hs = HamStage2(faces, V, 48, 512, 512, sc["weights"], edge_length_mean(vt, faces))
delta = torch.zeros(V, 3, device=dev, requires_grad=True)
albedo = sc["albedo0"][None].to(dev).clone().requires_grad_(True)
opt = Adam([{'params': delta, 'lr': sc["lr"]["lr"]}, {'params': albedo, 'lr': sc["lr"]["albedo_lr"]}])
for it in range(int(__import__("os").environ.get("FMHR_WTS_STEPS", "60"))):
    hs.step(vt, delta.detach(), albedo.detach()[0], sh, w2c, proj, tg["img"], tg["mask"], tg["valid_mask"], optimizer=opt, params=(delta, albedo))
torch.cuda.synchronize()


def report(name, fn, nwaves, slots_per_simd):
    buf = np.zeros(81920 * 4, dtype=np.uint64)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(-1, 4)[:nwaves].astype(np.int64)
    np.save(f"gpurun_out/wts_{name}.npy", t)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    life = end - start
    T = end.max()
    hw, xcc = t[:, 2] & 0xffffffff, (t[:, 2] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 7) << 4)
    slots = 256 * 4 * slots_per_simd
    print(f"== {name}: {nwaves} waves, span {T:.1f} us; wave life mean {life.mean():.2f} median {np.median(life):.2f} p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f} us")
    print(f"   sum of lives {life.sum() / 1e3:.1f} ms.wave = {life.sum() / T:.0f} resident on average of {slots} slots ({100 * life.sum() / T / slots:.0f} %)")
    for q in (0.5, 0.9, 0.95, 0.99):
        print(f"   {100 * q:.0f} % of the waves have ended by {np.quantile(end, q):.1f} us", end=";")
    print()
    print("   resident waves at 10 % steps of the span:", [int(((start <= x) & (end > x)).sum()) for x in np.linspace(0.05, 0.95, 10) * T])
    for x in range(8):
        m = xcc == x
        if m.any(): print(f"   XCD {x}: waves {m.sum()}, wave-time {life[m].sum():.0f} us, last end {end[m].max():.1f} us")
    key = xcc * 256 + cu
    u = np.unique(key)
    busy = np.array([life[key == k].sum() for k in u])
    last = np.array([end[key == k].max() for k in u])
    print(f"   per CU: wave-time min/mean/max {busy.min():.0f}/{busy.mean():.0f}/{busy.max():.0f} us; last end min/mean/max {last.min():.1f}/{last.mean():.1f}/{last.max():.1f} us")


report("tris", hs.lib.fmhr_debug_wts_raster, ((F + 255) // 256) * 48 * 4, 8)
report("face_bwd", hs.lib.fmhr_debug_wts_shade, ((F + 31) // 32) * 4 * 2, 4)     # both view groups (32 + 16 views)
tf = np.load("gpurun_out/wts_face_bwd.npy"); nb = ((F + 31) // 32) * 4
for gi, nm in ((0, "32-view group"), (1, "16-view group")):
    q = tf[gi * nb:(gi + 1) * nb]; t0f = tf[:, 0].min()
    st, en = (q[:, 0] - t0f) / 100.0, (q[:, 1] - t0f) / 100.0
    print(f"   face_bwd {nm}: starts {st.min():.1f} .. {np.median(st):.1f} .. {st.max():.1f} us, ends median {np.median(en):.1f} p99 {np.percentile(en, 99):.1f} max {en.max():.1f} us, life median {np.median(en - st):.2f} p90 {np.percentile(en - st, 90):.2f} sum {((en - st).sum() / 1e3):.1f} ms.wave")
report("aa_fwd_list", hs.lib.fmhr_debug_wts_aaf, 768 * 4, 3)
report("aa_bwd_items", hs.lib.fmhr_debug_wts_aab, 768 * 4, 7)
nvb = (V * 8 + 255) // 256
report("mesh_tail_a", hs.lib.fmhr_debug_wts_taila, nvb * 4, 8)
report("mesh_tail_b", hs.lib.fmhr_debug_wts_tailb, (nvb + 1) * 4, 8)
tb = np.load("gpurun_out/wts_mesh_tail_b.npy"); print("   tail_b finish block (workgroup 0): waves end at", [(round((x - tb[:, 0].min()) / 100.0, 2)) for x in tb[:4, 1]], "us")

t = np.load("gpurun_out/wts_aa_fwd_list.npy")
ph = t[:, 3].astype(np.uint64)
m1, m2, m3 = (ph & np.uint64(0xfffff)).astype(float) / 100.0, ((ph >> np.uint64(20)) & np.uint64(0xfffff)).astype(float) / 100.0, ((ph >> np.uint64(40)) & np.uint64(0xfff)).astype(float) / 100.0
nc = (ph >> np.uint64(52)).astype(int)
life = (t[:, 1] - t[:, 0]) / 100.0
print(f"aa_fwd_list per wave (median / p90): scan done {np.median(m1):.2f} / {np.percentile(m1, 90):.2f} us, past the barrier {np.median(m2):.2f} / {np.percentile(m2, 90):.2f}, "
      f"candidates analysed {np.median(m3):.2f} / {np.percentile(m3, 90):.2f} (12-bit wrap at 40.96), wave end {np.median(life):.2f} / {np.percentile(life, 90):.2f}; "
      f"candidate pairs per workgroup median {np.median(nc[::4])}, max {nc.max()}, total {nc[::4].sum()}")

px = t[:, 2].astype(np.uint64)
f = lambda k: ((px >> np.uint64(16 * k)) & np.uint64(0xffff)).astype(float) / 100.0
a_, b_, c_, d_ = f(0), f(1), f(2), f(3)
ok = d_ > 0
print(f"aa_fwd_list scan phases from wave start (median over waves with candidates): prologue done {np.median(a_[ok]):.2f} us, mask words back {np.median(b_[ok]):.2f}, "
      f"candidates listed {np.median(c_[ok]):.2f}, their pixel words back {np.median(d_[ok]):.2f}, pre-test done {np.median(m1[ok]):.2f}")
