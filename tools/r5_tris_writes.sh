#!/bin/bash
# which stores of k_raster_tris are its 116 MB of WRITE_SIZE (48 MB of 8-B entries + coverage entries)?  Measurement
# builds that drop ONE store class each (build/ab/{full,nosilh,nocov,noentry}.so), rocprofv3 --pmc WRITE_SIZE (+ the 32-B / 64-B write
# request counters where the pool has them) over a short bench run; the kernel's mean per launch.
#   gpurun --timeout 900 -- 'bash tools/r5_tris_writes.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cp $R/fmhr_amd/libfmhr_ham.so /tmp/keep.so
rocprofv3 --list-avail 2>/dev/null | grep -o -E "TCC_EA0_WRREQ[A-Za-z0-9_]*|TCC_EA0_WR_UNCACHED[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/r05_tris_writes.txt; echo >> $R/gpurun_out/r05_tris_writes.txt
for v in full nosilh nocov noentry; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  for C in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    rm -rf /tmp/tw
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/tw -o pmc --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --preroll 128 --no-cpu-baseline --no-stage1 --no-run-length > /tmp/tw.log 2>&1 || { echo "$v $C: pass failed"; tail -3 /tmp/tw.log; continue; }
    python3 - "$v" <<'PY' >> $R/gpurun_out/r05_tris_writes.txt
import csv, glob, sys
from collections import defaultdict
f = glob.glob("/tmp/tw/**/*counter_collection.csv", recursive=True)[0]
acc, n = defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(f)):
    if "k_raster_tris" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print(sys.argv[1].ljust(8), "  ".join("%s %.0f per launch (%d launches)" % (k, acc[k] / max(len(n[k]), 1), len(n[k])) for k in sorted(acc)))
PY
  done
done
cp /tmp/keep.so $R/fmhr_amd/libfmhr_ham.so
cat $R/gpurun_out/r05_tris_writes.txt
