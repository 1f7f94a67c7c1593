#!/bin/bash
# kernel statistics of the import-swap (op-by-op) iteration: which kernels its ~10 ms go to
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/obo_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/obo_prof -- python3 $R/tools/opbyop_rate.py > $R/gpurun_out/obo.log 2>&1
tail -1 $R/gpurun_out/obo.log
f=$(find $R/gpurun_out/obo_prof -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per iteration (7 iterations): %.2f ms" % (tot / 7e6))
for r in rows[:32]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), "%8.1f us avg" % (float(r["AverageNs"]) / 1e3), "%6.2f ms/iter" % (float(r["TotalDurationNs"]) / 7e6))
PY
