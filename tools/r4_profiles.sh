#!/bin/bash
# Round-4 profile set of the committed library for ONE configuration: rocprofv3 kernel stats of the bench command, the two
# HBM-traffic passes (FETCH_SIZE / WRITE_SIZE, separate), the issue-counter passes.
#   TAG=r04_v92 tools/r4_profiles.sh                       (metric config)
#   TAG=r04_v92_C4 BENCH_ARGS="--config C4" tools/r4_profiles.sh ;  TAG=r04_v92_6views BENCH_ARGS="--views 6" ...
# Outputs under gpurun_out/ (copied into profiles/ by hand).  ISSUE=0 skips the issue-counter passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r04_vXX}
export BENCH_ARGS
mkdir -p $R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham ${BENCH_ARGS:-} > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_p.log ) || { tail -20 $R/gpurun_out/${TAG}_p.log; exit 1; }
cp $(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -14 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
# the same trace over the timed region only (the last STEPS launches of every step kernel): what the bench line's HIP events saw
python3 $R/tools/trace_window.py $(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1) ${STEPS:-200} $R/gpurun_out/${TAG}_per_launch.csv > $R/gpurun_out/${TAG}_kernel_stats_timed.csv && cat $R/gpurun_out/${TAG}_kernel_stats_timed.csv
bash $R/tools/pmc_traffic.sh $TAG | tail -16 || exit 1
[ "${ISSUE:-1}" = "0" ] || python3 $R/tools/pmc_issue.py $TAG | tail -8
# the raw rocprofv3 directories (per-dispatch CSVs, tens of MB) are scratch: only the summaries above travel back (gpurun merges <= 64 MiB)
rm -rf $R/gpurun_out/$TAG $R/gpurun_out/${TAG}_FETCH_SIZE $R/gpurun_out/${TAG}_WRITE_SIZE $R/gpurun_out/${TAG}_issue[0-9]

