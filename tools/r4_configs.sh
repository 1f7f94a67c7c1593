#!/bin/bash
# every BASELINE.json configuration on one GPU with the committed library: gpurun_out/<TAG>_bench_<cfg>.json
#   gpurun --timeout 900 -- 'TAG=v93 bash tools/r4_configs.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-vXX}
mkdir -p $R/gpurun_out
for cfg in ${CONFIGS:-C1 C2 C3 C4 C5}; do
  python $R/bench.py --config $cfg --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/cfg_$cfg.log 2>&1 || { tail -5 $R/gpurun_out/cfg_$cfg.log; exit 1; }
  tail -1 $R/gpurun_out/cfg_$cfg.log > $R/gpurun_out/${TAG}_bench_$cfg.json
  python -c "
import json; d = json.load(open('$R/gpurun_out/${TAG}_bench_$cfg.json')); k = d['kernels_ms']
print('$cfg', round(d['value'], 1), 'it/s', round(d['ms_per_step'], 4), 'ms |', d['config']['rasterizer'], '|', {x: round(1e3 * v, 1) for x, v in k.items() if v > 0.006})"
done
