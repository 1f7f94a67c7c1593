#!/usr/bin/env python3
"""Per-kernel duration statistics over the LAST `steps` launches of every step kernel in a rocprofv3 kernel trace — the timed region of
`bench.py --steps K` — next to the whole-run figures `--stats` prints (which average the untimed pre-roll and warm-up steps in: the
triangle kernel is ~4 us shorter at the start of the optimisation than in the stationary window the HIP events of the bench line see).
    python3 tools/trace_window.py <kernel_trace.csv> <steps> [per_launch.csv] > profiles/<tag>_kernel_stats_timed.csv
The optional third argument receives every launch of the step kernels (kernel, launch index, start since the first launch in us, duration
in ns) — small enough to travel back from the GPU box, unlike the trace itself.
"""
import csv
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    runs = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "fmhr::" not in name:
                continue
            short = name.split("fmhr::", 1)[1].split("(", 1)[0]
            runs[short].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "CallsInWindow", "AverageNs", "MinNs", "MaxNs", "WholeRunCalls", "WholeRunAverageNs"])
    rows = []
    for name, se in runs.items():
        if len(se) < steps:                                     # not a per-step kernel (plan building, densify ...)
            continue
        se.sort()
        d = [e - s for s, e in se]
        win = d[-steps:]
        rows.append((sum(win) / len(win), name, len(win), min(win), max(win), len(d), sum(d) / len(d)))
    for avg, name, k, lo, hi, calls, whole in sorted(rows, reverse=True):
        w.writerow([name, k, round(avg, 1), lo, hi, calls, round(whole, 1)])
    w.writerow(["(sum of the step's kernels)", "", round(sum(r[0] for r in rows), 1), "", "", "", round(sum(r[6] for r in rows), 1)])
    if len(sys.argv) > 3:
        t0 = min(se[0][0] for se in runs.values() if len(se) >= steps)
        with open(sys.argv[3], "w", newline="") as f:
            pl = csv.writer(f)
            pl.writerow(["Name", "Launch", "StartUs", "DurationNs"])
            for name, se in sorted(runs.items()):
                if len(se) >= steps:
                    for i, (s, e) in enumerate(sorted(se)):
                        pl.writerow([name, i, round((s - t0) / 1e3, 1), e - s])


if __name__ == "__main__":
    main()
