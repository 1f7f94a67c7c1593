#!/bin/bash
# rocprofv3 kernel statistics of one bench run over its timed region: gpurun -- 'ENVS="FMHR_BINS_LIST=1" TAG=x BENCH_ARGS="..." bash tools/kstats.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-kstats}
mkdir -p $R/gpurun_out
for e in ${ENVS:-}; do export $e; done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-stage1 --no-run-length ${BENCH_ARGS:-} > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_p.log ) || { tail -20 $R/gpurun_out/${TAG}_p.log; exit 1; }
python3 $R/tools/trace_window.py $(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1) ${STEPS:-100} /dev/null | tee $R/gpurun_out/${TAG}_kernel_stats_timed.csv
rm -rf $R/gpurun_out/$TAG
