#!/bin/bash
# HBM traffic of the fused stage-1 step's two kernels (rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE separately, --kernel-trace only):
#   gpurun --timeout 600 -- 'bash tools/pmc_traffic_stage1.sh r02_stage1'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-stage1}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_$C -o pmc --output-format csv -- python3 $R/tools/stage1_rate.py 48 fusedonly > $R/gpurun_out/${TAG}_$C.log 2>&1 || { tail -20 $R/gpurun_out/${TAG}_$C.log; exit 1; }
  echo "pass $C done"
done
python3 - <<PY > $R/gpurun_out/${TAG}_hbm_traffic.txt
import csv, glob
from collections import defaultdict
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/${TAG}_" + c + "/**/*counter_collection.csv", recursive=True)[0]
    acc, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "k_s1_fwd" in r["Kernel_Name"] or "k_s1_bwd" in r["Kernel_Name"]:
            k = "k_s1_fwd" if "k_s1_fwd" in r["Kernel_Name"] else "k_s1_bwd"
            acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    out[c] = {k: acc[k] / len(disp[k]) for k in acc}
print("# fused stage-1 step, 48 views x 512x512 (tools/stage1_rate.py 48 fusedonly); mean KB per launch; gfx950: FETCH_SIZE counts 64 B per 128-B request on coalesced streams (x2 = upper estimate)")
print("kernel        FETCH_KB   WRITE_KB   HBM MB (fetch x1 .. x2 + write)")
for k in ("k_s1_fwd", "k_s1_bwd"):
    fe, wr = out["FETCH_SIZE"].get(k, 0), out["WRITE_SIZE"].get(k, 0)
    print(f"{k:12s} {fe:9.0f} {wr:10.0f}   {(fe + wr) * 1024 / 1e6:.0f} .. {(2 * fe + wr) * 1024 / 1e6:.0f}")
PY
cat $R/gpurun_out/${TAG}_hbm_traffic.txt
