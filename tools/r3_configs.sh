#!/bin/bash
# the other configurations of BASELINE.json on one GPU (bench.py --config; exact meshes, stationary) + the issue counters of the step
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for c in C1 C2 C3 C4 C5; do
  timeout -k 10 280 python $R/bench.py --config $c --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/cfg_$c.log 2>&1 || { tail -5 $R/gpurun_out/cfg_$c.log; exit 1; }
  tail -1 $R/gpurun_out/cfg_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$c', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', c['views'], 'x', c['H'], 'x', c['W'], 'V', c['V'], 'frames', c.get('frames'), c['rasterizer'], 'aa', c['aa_items_per_step'], 'preroll', c['preroll_steps'])"
done
