#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for s in 5 0; do
  rm -rf $R/gpurun_out/floor_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/floor_$s -- python3 $R/tools/analysis/empty_floor.py $s > $R/gpurun_out/floor_$s.log 2>&1
  f=$(find $R/gpurun_out/floor_$s -name "*kernel_stats.csv" | head -1)
  echo "== shift $s"; python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
done
