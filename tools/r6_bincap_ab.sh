#!/bin/bash
# does the bins' capacity (512 default since round 6, 1024 before) move the kernels?  same library, FMHR_BIN_CAP, two alternating rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
for cfg in "" "--config C4"; do for rep in 1 2; do for cap in 512 1024; do
  FMHR_BIN_CAP=$cap python $R/bench.py $cfg --steps 100 --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham > $O/cap.log 2>&1 || { tail -5 $O/cap.log; exit 1; }
  tail -1 $O/cap.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$cfg cap=$cap'.ljust(24), round(d['value'],1), {x: round(k[x]*1e3,1) for x in ('raster_tris','raster_resolve','shade_bwd')})"
done; done; done
