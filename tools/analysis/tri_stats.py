#!/usr/bin/env python3
"""Workload statistics of the triangle kernel on the metric scene (CPU, numpy): per (view, triangle) box sizes and coverage
counts, and the per-WAVE quantities that set the kernel's trip counts (a wave = 64 consecutive faces of one view)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fmhr_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = synthetic.scene_from_config(cfg, 0)
H, W = sc["H"], sc["W"]
v = sc["vertices"].numpy().astype(np.float32); f = sc["faces"].numpy()
F = f.shape[0]
tot = {}
def acc(k, x): tot.setdefault(k, []).append(x)
for b in range(0, sc["views"], max(1, sc["views"] // nv)):
    cm = (sc["w2c"][b].numpy() @ sc["proj"][b].numpy()).astype(np.float32)
    p = np.concatenate([v, np.ones((v.shape[0], 1), np.float32)], 1) @ cm
    sx = np.rint(p[:, 0] / p[:, 3] * (W * 128) + W * 128).astype(np.int64)
    sy = np.rint(p[:, 1] / p[:, 3] * (H * 128) + H * 128).astype(np.int64)
    x = sx[f]; y = sy[f]
    pxa = np.maximum((x.min(1) - 128 + 255) >> 8, 0); pxb = np.minimum((x.max(1) - 128) >> 8, W - 1)
    pya = np.maximum((y.min(1) - 128 + 255) >> 8, 0); pyb = np.minimum((y.max(1) - 128) >> 8, H - 1)
    bw = pxb - pxa + 1; bh = pyb - pya + 1
    area2 = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (y[:, 1] - y[:, 0]) * (x[:, 2] - x[:, 0])
    ok = (bw > 0) & (bh > 0) & (area2 != 0)
    box = np.where(ok, bw * bh, 0)
    # exact coverage count per triangle (vectorised over the up-to-64 box pixels)
    cnt = np.zeros(F, np.int64)
    idx = np.nonzero(ok & (box <= 64))[0]
    xs = x[idx].copy(); ys = y[idx].copy()
    neg = area2[idx] < 0
    xs[neg, 1], xs[neg, 2] = xs[neg, 2].copy(), xs[neg, 1].copy(); ys[neg, 1], ys[neg, 2] = ys[neg, 2].copy(), ys[neg, 1].copy()
    for k in range(64):
        live = k < box[idx]
        c = k % np.maximum(bw[idx], 1); r = k // np.maximum(bw[idx], 1)
        cx = (pxa[idx] + c) * 256 + 128; cy = (pya[idx] + r) * 256 + 128
        ins = live.copy()
        for e in range(3):
            ax, ay, bx, by = xs[:, e], ys[:, e], xs[:, (e + 1) % 3], ys[:, (e + 1) % 3]
            dx, dy = bx - ax, by - ay
            E = dx * (cy - ay) - dy * (cx - ax)
            ins &= (E > 0) | ((E == 0) & ((dy > 0) | ((dy == 0) & (dx > 0))))
        cnt[idx] += ins
    nw = F // 64
    bx64 = box[:nw * 64].reshape(nw, 64); c64 = cnt[:nw * 64].reshape(nw, 64)
    acc("frac_box", ok.mean()); acc("frac_cov", (cnt > 0).mean()); acc("frags_per_tri", cnt.sum() / F)
    acc("mean_box", box[ok].mean()); acc("large", (box > 64).sum())
    acc("wave_max_box", bx64.max(1).mean()); acc("wave_max_cnt", c64.max(1).mean()); acc("wave_T", c64.sum(1).mean())
    acc("wave_nz", (c64 > 0).sum(1).mean()); acc("wave_anybox", (bx64 > 0).sum(1).mean())
    acc("wave_T0", (c64.sum(1) == 0).mean()); acc("wave_rounds64", np.ceil(c64.sum(1) / 64).mean())
    acc("wave_sum_box", bx64.sum(1).mean())
    hist = np.bincount(np.minimum(box[ok], 65), minlength=66)
    acc("box_hist", hist / ok.sum())
    acc("wmb_hist", np.bincount(np.minimum(bx64.max(1), 65), minlength=66) / nw)
for k, vs in tot.items():
    if k.endswith("hist"):
        h = np.mean(vs, 0); cum = np.cumsum(h)
        print(k, " ".join(f"{i}:{cum[i]:.2f}" for i in (1, 2, 3, 4, 6, 8, 9, 12, 16, 20, 25, 32, 48, 64, 65)))
    else:
        print(f"{k:16s} {np.mean(vs):.3f}")
