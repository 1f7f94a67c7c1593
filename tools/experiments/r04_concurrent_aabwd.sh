#!/bin/bash
# EXPERIMENT (round 4, not kept): how much of k_aa_bwd_items (17 us, a chain of ~8 dependent round trips at 10 % occupancy) hides under
# k_loss_pix (28 us, bandwidth-bound) when both are resident?  build/ab/conc.so = the library with tools/experiments/
# r04_concurrent_aabwd.patch applied and -DFMHR_X_CONCURRENT_AABWD (the antialias backward on a side stream next to k_loss_pix: results
# wrong — both write g_shaded — timing valid), build/ab/base.so = the committed library.  From the kernel trace of the last 30 steps:
#   base  3024 it/s | first start of (loss, aa_bwd) -> face_bwd start: 45.5 us | loss 28.7 us, aa_bwd 16.6 us
#   conc  2856 it/s | first start of (loss, aa_bwd) -> face_bwd start: 52.4 us | loss 29.0 us, aa_bwd 32.5 us
# Next to a kernel that saturates the memory system every one of the antialias backward's round trips takes twice as long: the pair would
# take ~33 us instead of 45 in ONE launch (no stream hand-overs) at best — for a second gradient-ownership protocol on g_shaded.  Not built.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for v in base conc; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_$v -o tr_$v --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --preroll 128 --no-cpu-baseline --no-stage1 --no-run-length > $O/tr_$v.json 2> $O/tr_$v.log ) || { tail -20 $O/tr_$v.log; exit 1; }
  python3 - <<PY
import csv, glob, json
f = glob.glob("$O/tr_$v/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fmhr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(r): return r["Kernel_Name"].split("fmhr::",1)[1].split("(",1)[0][:14]
# last 30 steps: span from k_loss_pix start to k_face_bwd start
idx = [i for i, r in enumerate(rows) if short(r).startswith("k_loss_pix")][-30:]
spans, lo, aa = [], [], []
for i in idx:
    j = next(k for k in range(i - 3, i + 6) if short(rows[k]).startswith("k_face_bwd"))
    a = next(k for k in range(max(i - 3, 0), i + 6) if short(rows[k]).startswith("k_aa_bwd"))
    s0 = min(int(rows[i]["Start_Timestamp"]), int(rows[a]["Start_Timestamp"]))
    spans.append(int(rows[j]["Start_Timestamp"]) - s0)
    lo.append(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])); aa.append(int(rows[a]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
d = json.loads(open("$O/tr_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), "it/s | first start of (loss, aa_bwd) -> face_bwd start: %.1f us | loss %.1f us, aa_bwd %.1f us" % (sum(spans)/len(spans)/1e3, sum(lo)/len(lo)/1e3, sum(aa)/len(aa)/1e3))
PY
  rm -rf $O/tr_$v
done
