#!/bin/bash
# Round-4 GPU cycle: the whole GPU suite with the parity log (gpurun_out/parity_r04.txt), then the driver-form bench (--steps 20
# --warmup 5) and a 200-step run.   gpurun --timeout 1100 -- 'bash tools/r4_cycle.sh'       (PYTEST_ARGS="-k pitch" for a subset; BENCH=0 skips the bench)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/parity_r04.txt
timeout -k 10 ${TEST_TIMEOUT:-800} python -m pytest $R/tests ${PYTEST_X--x} -q -m gpu ${PYTEST_ARGS:-} > $R/gpurun_out/t.log 2>&1 || { tail -60 $R/gpurun_out/t.log; exit 1; }
tail -1 $R/gpurun_out/t.log
[ "${BENCH:-1}" = "0" ] && exit 0
timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/b20.log 2>&1 || { tail -30 $R/gpurun_out/b20.log; exit 1; }
timeout -k 10 300 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/b200.log 2>&1 || { tail -30 $R/gpurun_out/b200.log; exit 1; }
python - <<PY
import json
for f in ("b20", "b200"):
    d = json.loads(open("$R/gpurun_out/%s.log" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 4), d["config"]["preroll_steps"], d["config"]["aa_items_per_step"], d["config"]["V"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
    print({k: round(v, 4) for k, v in d["kernels_ms"].items() if v > 0.001})
PY
