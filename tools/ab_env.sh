#!/bin/bash
# A/B of ONE library under an environment switch on ONE box: ENVVAR=FMHR_RASTER_BINNED VALUES="0 1" KEYS=raster_tris,raster_resolve
#   gpurun --timeout 900 -- 'ENVVAR=FMHR_RASTER_BINNED VALUES="0 1" KEYS=raster_tris,raster_resolve bash tools/ab_env.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for rep in $(seq 1 ${REPS:-2}); do for v in ${VALUES:-0 1}; do
  env ${ENVVAR}=$v python $R/bench.py --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham ${BENCH_ARGS:-} > $R/gpurun_out/abe_$v.log 2>&1 || { tail -5 $R/gpurun_out/abe_$v.log; exit 1; }
  tail -1 $R/gpurun_out/abe_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('${ENVVAR}=$v'.ljust(24), round(d['value'],1), round(d['ms_per_step'],4), {x:round(k[x],4) for x in ('${KEYS:-regularisers}'.split(','))})"
done; done
