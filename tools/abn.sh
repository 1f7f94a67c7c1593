#!/bin/bash
# A/B/C... of several prebuilt libraries on ONE box: VARIANTS="old c2 c4" (files build/ab/<name>.so), two alternating rounds.
#   gpurun --timeout 900 -- 'VARIANTS="old c2 c4" KEYS=raster_tris,shade_bwd bash tools/abn.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in ${VARIANTS:-old new}; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  python $R/bench.py --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham ${BENCH_ARGS:-} > $R/gpurun_out/ab_$v.log 2>&1 || { tail -5 $R/gpurun_out/ab_$v.log; exit 1; }
  tail -1 $R/gpurun_out/ab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v'.ljust(8), round(d['value'],1), {x:round(k[x],4) for x in ('${KEYS:-regularisers}'.split(','))})"
done; done
