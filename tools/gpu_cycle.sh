#!/bin/bash
# One GPU-box cycle: parity tests, then the default bench, then (optionally, PROF=name) a rocprofv3 kernel-stats pass.
# Run through gpurun from the repo root:  gpurun --timeout 600 -- 'bash tools/gpu_cycle.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 400 python -m pytest $R/tests -x -q -m gpu > $R/gpurun_out/t.log 2>&1 || { tail -30 $R/gpurun_out/t.log; exit 1; }
tail -1 $R/gpurun_out/t.log
timeout -k 10 200 python $R/bench.py --steps ${STEPS:-200} --warmup 20 ${BENCH_ARGS:-} > $R/gpurun_out/b.log 2>&1 || { tail -30 $R/gpurun_out/b.log; exit 1; }
python - <<PY
import json
d = json.loads(open("$R/gpurun_out/b.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("step_roofline", {}).get("frac"))
print(d["kernels_ms"])
PY
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$PROF -o $PROF --output-format csv -- python3 $R/bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/p.log 2>&1 || { tail -20 $R/gpurun_out/p.log; exit 1; }
  python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/$PROF/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:${TOP:-24}]:
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(4), round(float(r["AverageNs"]) / 1000, 1))
PY
fi
