#!/bin/bash
# Round-6 profile set of the committed library: metric config (kernel stats over the timed region, HBM traffic, issue counters) and C4.
#   gpurun --timeout 1180 -- 'V=v104 bash tools/r6_final_profiles.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${V:-v104}
STEPS=200 TAG=r06_${V} BENCH_ARGS="" bash $R/tools/r4_profiles.sh > $R/gpurun_out/r06_${V}.out 2>&1 || { tail -20 $R/gpurun_out/r06_${V}.out; exit 1; }
ISSUE=0 STEPS=200 TAG=r06_${V}_C4 BENCH_ARGS="--config C4" bash $R/tools/r4_profiles.sh > $R/gpurun_out/r06_${V}_C4.out 2>&1 || { tail -20 $R/gpurun_out/r06_${V}_C4.out; exit 1; }
python3 $R/tools/traffic_json.py $R/gpurun_out/r06_${V}_hbm_traffic.txt $R/gpurun_out/traffic_r06.json
cat $R/gpurun_out/r06_${V}_kernel_stats_timed.csv $R/gpurun_out/r06_${V}_C4_kernel_stats_timed.csv
for c in C1 C2 C3 C4 C5; do python3 $R/bench.py --config $c --no-stage1 --no-run-length --no-cpu-baseline > $R/gpurun_out/r06_${V}_bench_$c.json 2> $R/gpurun_out/r06_${V}_bench_$c.err || { tail -5 $R/gpurun_out/r06_${V}_bench_$c.err; exit 1; }; python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/r06_${V}_bench_$c.json').read().strip().splitlines()[-1]); print('$c', round(d['value'],1), d['config']['bin_depth_max'], d['config']['bin_cap'], d['config']['bin_cap_grown'])"; done
