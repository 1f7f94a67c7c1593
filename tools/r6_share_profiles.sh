#!/bin/bash
# the per-rank share the 8-GPU runs execute — kernel statistics over the timed region and the HBM-traffic passes for
#   (a) ONE view x 1280 x 1024 of the two-hand mesh (the demo conf's batch of 8 on 8 GPUs)   (b) 6 views x 1024 x 1024 (C4 on 8 GPUs)
#   gpurun --timeout 1100 -- 'V=v104 bash tools/r6_share_profiles.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${V:-v104}
ISSUE=0 STEPS=200 TAG=r06_${V}_1view_demo BENCH_ARGS="--config demo_batch --views 1" bash $R/tools/r4_profiles.sh > $R/gpurun_out/r06_${V}_1view_demo.out 2>&1 || { tail -20 $R/gpurun_out/r06_${V}_1view_demo.out; exit 1; }
ISSUE=0 STEPS=200 TAG=r06_${V}_6views_C4 BENCH_ARGS="--config C4 --views 6" bash $R/tools/r4_profiles.sh > $R/gpurun_out/r06_${V}_6views_C4.out 2>&1 || { tail -20 $R/gpurun_out/r06_${V}_6views_C4.out; exit 1; }
cat $R/gpurun_out/r06_${V}_1view_demo_kernel_stats_timed.csv $R/gpurun_out/r06_${V}_6views_C4_kernel_stats_timed.csv
