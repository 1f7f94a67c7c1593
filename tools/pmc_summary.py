#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into mean KB per launch and kernel."""
import csv
import glob
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    acc, cnt = defaultdict(float), defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter:
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        acc[r["Kernel_Name"]] += float(r["Counter_Value"])
        if key not in seen:
            seen.add(key)
            cnt[r["Kernel_Name"]] += 1
    return acc, cnt


fa, fc = load(sys.argv[1], "FETCH_SIZE")
wa, wc = load(sys.argv[2], "WRITE_SIZE")
names = sorted(set(fa) | set(wa), key=lambda k: -(fa.get(k, 0) + wa.get(k, 0)))
print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes: they cannot share one)")
print("# command: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage1 --no-run-length")
print("# metric config: 48 views x 512x512, V=49281, F=98432 (pre-rolled to the stationary antialias list).  Values: mean KB per launch.")
print("# gfx950 caveat: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams (x2 correction = the 'high' total).")
print(f"{'kernel':44s} {'launches':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s}")
tot_f = tot_w = 0.0
steps = None
for k in names:
    nl = max(fc.get(k, 0), wc.get(k, 0), 1)
    fk, wk = fa.get(k, 0) / max(fc.get(k, 1), 1), wa.get(k, 0) / max(wc.get(k, 1), 1)
    print(f"{k[:44]:44s} {nl:8d} {fk:12.0f} {wk:12.0f}")
    if "k_face_bwd" in k:
        steps = nl
for k in names:
    if steps and k.startswith(("fmhr::", "void fmhr::")):
        per_step = max(fc.get(k, 0), wc.get(k, 0)) / steps
        if per_step >= 0.9:                                   # kernels of the step (setup kernels run once)
            tot_f += fa.get(k, 0) / steps
            tot_w += wa.get(k, 0) / steps
print(f"per-step total over the step's kernels: FETCH {tot_f * 1024 / 1e9:.3f} GB raw ({2 * tot_f * 1024 / 1e9:.3f} GB with the x2 stream "
      f"correction), WRITE {tot_w * 1024 / 1e9:.3f} GB")
print(f"=> measured HBM traffic per step {(tot_f + tot_w) * 1024 / 1e9:.2f} - {(2 * tot_f + tot_w) * 1024 / 1e9:.2f} GB")
