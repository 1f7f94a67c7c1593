#!/bin/bash
# Issue / wait profile of the step's kernels: SQ counters in two rocprofv3 --pmc passes (kernel-trace only).
#   gpurun --timeout 900 -- 'bash tools/pmc_sq.sh r01_v54'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
P=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  P=$((P+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_sq$P -o pmc --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/${TAG}_sq$P.log 2>&1 || { tail -20 $R/gpurun_out/${TAG}_sq$P.log; exit 1; }
  echo "pass $P done"
done
python3 - <<PY > $R/gpurun_out/${TAG}_sq_counters.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for p in (1, 2):
    f = glob.glob("$R/gpurun_out/${TAG}_sq%d/**/*counter_collection.csv" % p, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not ("fmhr::" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"]
print("# rocprofv3 --kernel-trace --pmc (two passes), bench.py --steps 10 --warmup 2: mean per launch")
print("kernel".ljust(36), " ".join(n[3:].rjust(16) for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    nl = max(1, len(cnt[k]) // 2)
    print(k[:36].ljust(36), " ".join(("%.3g" % (acc[k].get(n, 0) / nl)).rjust(16) for n in names))
PY
head -16 $R/gpurun_out/${TAG}_sq_counters.txt
