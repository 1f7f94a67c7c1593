#!/bin/bash
# the per-rank shares of the 2- and 4-GPU runs of the metric (24 / 12 views x 512 x 512) and the demo loop on one GPU: kernel statistics over the timed region
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${V:-v104}
for spec in "24views:--views 24" "12views:--views 12" "demo_loop:--config demo"; do
  tag=${spec%%:*}; args=${spec#*:}
  STEPS=200 TAG=r06_${V}_${tag} BENCH_ARGS="$args --no-run-ham" bash $R/tools/kstats.sh > $R/gpurun_out/r06_${V}_${tag}.out 2>&1 || { tail -20 $R/gpurun_out/r06_${V}_${tag}.out; exit 1; }
  tail -1 $R/gpurun_out/r06_${V}_${tag}_kernel_stats_timed.csv
done
