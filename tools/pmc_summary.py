#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs of the same command) into mean KB per launch and kernel,
and the HBM traffic of ONE STEP.

    pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>          (raw counter_collection.csv of each pass)
    pmc_summary.py --retotal <an existing summary table>                              (recompute the per-step total from the table)

Round 5: the two passes pre-roll to different lengths (bench.py stops when the antialias list is stationary), so
they hold different numbers of launches.  Each pass is normalised by ITS OWN launch count: mean KB per launch = the pass's counter sum /
the pass's launches, and per-step total = sum over the step's kernels of (mean KB per launch x launches per step), launches per step
taken per pass as well.  (Rounds 1-4 divided each pass's sum by the OTHER pass's step count: the text total of
profiles/r04_v99_hbm_traffic.txt said WRITE 0.245 GB where its own table sums to 0.339 GB.)  The total printed here equals the sum of
the table's rows and what tools/traffic_json.py writes."""
import csv
import glob
import re
import sys
from collections import defaultdict

STEP_MARK = "k_face_bwd"                                  # launched exactly once per step: its count = the pass's number of steps


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


def is_step_kernel(name, per_step):
    return name.startswith(("fmhr::", "void fmhr::")) and per_step >= 0.9


def total_lines(rows):
    """rows: (name, per_step_f, per_step_w, fetch_kb_per_launch, write_kb_per_launch) of the step's kernels"""
    tot_f = sum(r[3] * r[1] for r in rows)
    tot_w = sum(r[4] * r[2] for r in rows)
    return [f"per-step total over the step's kernels (sum of the rows above x their launches per step): FETCH {tot_f * 1024 / 1e9:.3f} GB raw "
            f"({2 * tot_f * 1024 / 1e9:.3f} GB with the x2 stream correction), WRITE {tot_w * 1024 / 1e9:.3f} GB",
            f"=> measured HBM traffic per step {(tot_f + tot_w) * 1024 / 1e9:.3f} - {(2 * tot_f + tot_w) * 1024 / 1e9:.3f} GB"]


def retotal(path):
    """an existing table (any round's format): every kernel of the step launches once per step; the setup kernels (a handful of launches)
    are left out by their launch count"""
    lines = open(path).read().splitlines()
    rows, nmax = [], 0
    for ln in lines:
        m = re.match(r"^(\S.*?)\s+(\d+)(?:\s+(\d+)\s+([\d.]+))?\s+(\d+)\s+(\d+)\s*$", ln)
        if m and STEP_MARK in m.group(1):
            nmax = int(m.group(2))
    for ln in lines:
        m = re.match(r"^(\S.*?)\s+(\d+)(?:\s+(\d+)\s+([\d.]+))?\s+(\d+)\s+(\d+)\s*$", ln)
        if not m:
            continue
        name, nl = m.group(1), int(m.group(2))
        per = float(m.group(4)) if m.group(4) else (1.0 if nmax and nl >= 0.9 * nmax else 0.0)
        if is_step_kernel(name, per):
            rows.append((name, round(per), round(per), float(m.group(5)), float(m.group(6))))
    keep = [ln for ln in lines if not ln.startswith(("per-step total", "=> measured"))]
    print("\n".join(keep))
    print("# total recomputed in round 5 from the table above (tools/pmc_summary.py --retotal): the original total divided each pass's counter sum")
    print("# by the OTHER pass's launch count")
    print("\n".join(total_lines(rows)))


def main():
    if sys.argv[1] == "--retotal":
        return retotal(sys.argv[2])
    fa, fc = load(sys.argv[1], "FETCH_SIZE")
    wa, wc = load(sys.argv[2], "WRITE_SIZE")
    steps_f = max([v for k, v in fc.items() if STEP_MARK in k] or [0])
    steps_w = max([v for k, v in wc.items() if STEP_MARK in k] or [0])
    names = sorted(set(fa) | set(wa), key=lambda k: -(fa.get(k, 0) / max(fc.get(k, 1), 1) + wa.get(k, 0) / max(wc.get(k, 1), 1)))
    print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes: they cannot share one)")
    print("# command: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage1 --no-run-length " + " ".join(sys.argv[3:]))
    print(f"# steps in the FETCH_SIZE pass: {steps_f}, in the WRITE_SIZE pass: {steps_w} (each pass is normalised by its OWN launch counts).  Values: mean KB per launch.")
    print("# gfx950 caveat: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams (x2 correction = the 'high' total).")
    print(f"{'kernel':44s} {'launchF':>8s} {'launchW':>8s} {'per_step':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s}")
    rows = []
    for k in names:
        nf, nw = fc.get(k, 0), wc.get(k, 0)
        fk, wk = fa.get(k, 0) / max(nf, 1), wa.get(k, 0) / max(nw, 1)
        pf, pw = (nf / steps_f if steps_f else 0.0), (nw / steps_w if steps_w else 0.0)
        per = max(pf, pw)
        print(f"{k[:44]:44s} {nf:8d} {nw:8d} {per:8.2f} {fk:12.0f} {wk:12.0f}")
        if is_step_kernel(k, min(pf or per, pw or per)):
            rows.append((k, round(pf), round(pw), fk, wk))
    print("\n".join(total_lines(rows)))


if __name__ == "__main__":
    main()
