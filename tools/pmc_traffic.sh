#!/bin/bash
# HBM traffic per kernel of one bench step: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass,
# /opt/skills/guides/MI355X_MICROARCH.md) with --kernel-trace only.  Run through gpurun from the repo root:
#   gpurun --timeout 900 -- 'bash tools/pmc_traffic.sh r01_v45'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/${TAG}_$C -o pmc --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham ${BENCH_ARGS:-} > $R/gpurun_out/${TAG}_$C.log 2>&1 || { tail -20 $R/gpurun_out/${TAG}_$C.log; exit 1; }
  echo "pass $C done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_FETCH_SIZE $R/gpurun_out/${TAG}_WRITE_SIZE > $R/gpurun_out/${TAG}_hbm_traffic.txt
tail -40 $R/gpurun_out/${TAG}_hbm_traffic.txt
