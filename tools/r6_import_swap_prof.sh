#!/bin/bash
# rocprofv3 kernel trace of the import-swap (op-by-op) iteration — the reference's loop body on fmhr_amd.dr / fmhr_amd.utils
# with torch autograd (fmhr_amd/examples/loop_body.py; what bench.py's `import_swap` leg times) — split into the library's kernels (fmhr::*)
# and torch's own (at::native::*, the script's indexing / elementwise / einsum work).   gpurun -- 'bash tools/r6_import_swap_prof.sh [tag]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06_import_swap}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/obo_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/obo_prof -- python3 $R/tools/opbyop_rate.py > $R/gpurun_out/${TAG}.log 2>&1 || { tail -20 $R/gpurun_out/${TAG}.log; exit 1; }
tail -1 $R/gpurun_out/${TAG}.log
f=$(find $R/gpurun_out/obo_prof -name "*kernel_stats.csv" | head -1)
python3 - <<PY > $R/gpurun_out/${TAG}_kernel_stats.csv
import csv, sys
rows = list(csv.DictReader(open("$f")))
ITERS = 7.0                                    # tools/opbyop_rate.py: 2 warm-up + 5 timed iterations, 48 views x 512 x 512
def cls(n):
    # torch's own: ATen kernels (incl. the anonymous-namespace indexing backward), rocPRIM sorts / scans behind index_put, Tensile GEMMs (einsum);
    # "runtime": the HIP runtime's copy / fill kernels (torch.zeros, .cuda(), clones)
    if "fmhr::" in n: return "fmhr"
    if "__amd_rocclr" in n: return "runtime"
    return "torch" if ("at::native" in n or "at_cuda" in n or "rocprim" in n or "hipcub" in n or "Cijk" in n or "elementwise" in n or "indexing_backward" in n or "anonymous namespace" in n) else "other"
tot = {}
for r in rows:
    tot[cls(r["Name"])] = tot.get(cls(r["Name"]), 0.0) + float(r["TotalDurationNs"])
w = csv.writer(sys.stdout)
w.writerow(["class_or_kernel", "calls_per_iteration", "avg_us", "ms_per_iteration"])
for k in ("fmhr", "torch", "runtime", "other"):
    w.writerow(["(all %s kernels)" % k, "", "", round(tot.get(k, 0.0) / ITERS / 1e6, 3)])
w.writerow(["(all kernels)", "", "", round(sum(tot.values()) / ITERS / 1e6, 3)])
for r in rows[:60]:
    w.writerow([cls(r["Name"]) + ": " + r["Name"][:110], round(int(r["Calls"]) / ITERS, 1), round(float(r["AverageNs"]) / 1e3, 1), round(float(r["TotalDurationNs"]) / ITERS / 1e6, 3)])
PY
head -30 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-170
rm -rf $R/gpurun_out/obo_prof
