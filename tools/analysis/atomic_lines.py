#!/usr/bin/env python3
"""How many distinct 64-B lines of the depth-key buffer one atomic wave-instruction of k_raster_tris touches, for the real task
order (tasks of a wave = covered pixels of its 64 consecutive faces, face by face) and for alternatives.  CPU, numpy."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fmhr_amd import synthetic
sc = synthetic.scene_from_config("metric", 0)
H, W = sc["H"], sc["W"]
v = sc["vertices"].numpy().astype(np.float32); f = sc["faces"].numpy(); F = f.shape[0]
res = {}
def acc(k, x): res.setdefault(k, []).append(x)
for b in (0, 7, 19, 33, 41):
    cm = (sc["w2c"][b].numpy() @ sc["proj"][b].numpy()).astype(np.float32)
    p = np.concatenate([v, np.ones((v.shape[0], 1), np.float32)], 1) @ cm
    sx = np.rint(p[:, 0] / p[:, 3] * (W * 128) + W * 128).astype(np.int64)
    sy = np.rint(p[:, 1] / p[:, 3] * (H * 128) + H * 128).astype(np.int64)
    x = sx[f]; y = sy[f]
    pxa = np.maximum((x.min(1) - 128 + 255) >> 8, 0); pxb = np.minimum((x.max(1) - 128) >> 8, W - 1)
    pya = np.maximum((y.min(1) - 128 + 255) >> 8, 0); pyb = np.minimum((y.max(1) - 128) >> 8, H - 1)
    bw = pxb - pxa + 1; bh = pyb - pya + 1
    area2 = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (y[:, 1] - y[:, 0]) * (x[:, 2] - x[:, 0])
    ok = (bw > 0) & (bh > 0) & (area2 != 0) & (bw * bh <= 32)
    idx = np.nonzero(ok)[0]
    xs = x[idx].copy(); ys = y[idx].copy(); neg = area2[idx] < 0
    xs[neg, 1], xs[neg, 2] = xs[neg, 2].copy(), xs[neg, 1].copy(); ys[neg, 1], ys[neg, 2] = ys[neg, 2].copy(), ys[neg, 1].copy()
    frag_f, frag_x, frag_y = [], [], []
    for k in range(32):
        live = k < (bw * bh)[idx]
        c = k % np.maximum(bw[idx], 1); r = k // np.maximum(bw[idx], 1)
        px = pxa[idx] + c; py = pya[idx] + r
        cx = px * 256 + 128; cy = py * 256 + 128
        ins = live.copy()
        for e in range(3):
            ax, ay, bx, by = xs[:, e], ys[:, e], xs[:, (e + 1) % 3], ys[:, (e + 1) % 3]
            dx, dy = bx - ax, by - ay
            E = dx * (cy - ay) - dy * (cx - ax)
            ins &= (E > 0) | ((E == 0) & ((dy > 0) | ((dy == 0) & (dx > 0))))
        frag_f.append(idx[ins]); frag_x.append(px[ins]); frag_y.append(py[ins])
    ff = np.concatenate(frag_f); fx = np.concatenate(frag_x); fy = np.concatenate(frag_y)
    order = np.lexsort((fy * W + fx, ff))            # face-major, then row-major inside the box = the kernel's task order
    ff, fx, fy = ff[order], fx[order], fy[order]
    wave = ff // 64
    lines_row = fy * (W // 8) + fx // 8              # 64-B line of a row-major u64 key buffer
    lines_128 = fy * (W // 16) + fx // 16
    tile = (fy // 2) * (W // 4) + fx // 4            # 4 x 2 pixel tiles per 64-B line
    tile44 = (fy // 4) * (W // 4) + fx // 4          # 4 x 4 pixel tiles per 128-B line
    tot_instr = 0; tot = dict(row64=0, row128=0, tile42=0, tile44=0, sorted64=0, pool64=0)
    starts = np.flatnonzero(np.r_[True, wave[1:] != wave[:-1]]); ends = np.r_[starts[1:], len(wave)]
    for s, e in zip(starts, ends):
        for c0 in range(s, e, 64):
            c1 = min(c0 + 64, e); tot_instr += 1
            tot["row64"] += len(np.unique(lines_row[c0:c1])); tot["row128"] += len(np.unique(lines_128[c0:c1]))
            tot["tile42"] += len(np.unique(tile[c0:c1])); tot["tile44"] += len(np.unique(tile44[c0:c1]))
        tot["sorted64"] += len(np.unique(lines_row[s:e]))        # lower bound: every line of the wave's footprint once
    nw = len(starts)
    acc("atomic instr / wave", tot_instr / nw); acc("frags / wave", len(ff) / nw)
    for k, t in tot.items(): acc(f"lines/wave {k}", t / nw)
for k, vs in res.items(): print(f"{k:28s} {np.mean(vs):.2f}")
