#!/bin/bash
# the corner-slot skip of the few-views backward (FMHR_SLOT_SKIP=0 / 1, same library): tests, then it/s | shade_bwd us | the two tails us
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 700 python -m pytest $R/tests -x -q -m gpu > $O/t_skip.log 2>&1 || { tail -40 $O/t_skip.log; exit 1; }
tail -1 $O/t_skip.log
fi
for cfg in "--views 6" "--config demo_batch --views 1" "--config C4 --views 6" "--views 2" "--config C3 --views 8"; do
  for rep in 1 2; do for skip in 0 1; do
    FMHR_SLOT_SKIP=$skip python $R/bench.py $cfg --steps 100 --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham > $O/skip.log 2>&1 || { tail -5 $O/skip.log; exit 1; }
    tail -1 $O/skip.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$cfg skip=$skip'.ljust(44), round(d['value'],1), round(k['shade_bwd']*1e3,1), round(k['regularisers']*1e3,1))"
  done; done
done
