#!/bin/bash
# A/B of two prebuilt libraries on ONE box (boxes of the pool differ by +-3 %, more than most kernel changes):
#   cp fmhr_amd/libfmhr_ham.so build/ab/old.so; <edit, rebuild>; cp fmhr_amd/libfmhr_ham.so build/ab/new.so
#   gpurun --timeout 900 -- 'KEYS=aa_fwd,aa_bwd bash tools/ab.sh'          (BENCH_ARGS="--views 6" for the small-batch case)
# Alternates the two twice and prints it/s and the chosen launch groups (ms, HIP events of bench.py's profiling pass).
# build/ is git-ignored but travels to the box (only build/obj/ is in .gpurunignore).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in old new; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  python $R/bench.py --no-cpu-baseline --no-stage1 --no-run-length ${BENCH_ARGS:-} > $R/gpurun_out/ab_$v.log 2>&1 || { tail -5 $R/gpurun_out/ab_$v.log; exit 1; }
  tail -1 $R/gpurun_out/ab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$v', round(d['value'],1), {x:round(k[x],4) for x in ('${KEYS:-regularisers}'.split(','))})"
done; done
