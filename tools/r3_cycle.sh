#!/bin/bash
# Round-3 GPU cycle: (optional) validation that the stale-owner-mark regression test FAILS on a build without the wipe, the whole
# GPU suite with the parity log, then the default bench (driver form: --steps 20 --warmup 5) and a 200-step run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/parity_r03.txt
if [ -f $R/build/ab/nowipe.so ] && [ "${CHECK_NOWIPE:-0}" = "1" ]; then
  cp $R/fmhr_amd/libfmhr_ham.so $R/build/ab/_cur.so
  cp $R/build/ab/nowipe.so $R/fmhr_amd/libfmhr_ham.so
  FMHR_PARITY_LOG=$R/gpurun_out/parity_nowipe.txt timeout -k 10 200 python -m pytest $R/tests/test_gpu_ham.py -x -q -m gpu -k stale_owner > $R/gpurun_out/t_nowipe.log 2>&1
  echo "nowipe build: stale-owner test exit code $? (expected 1 = the test catches the defect)"; tail -3 $R/gpurun_out/t_nowipe.log
  cp $R/build/ab/_cur.so $R/fmhr_amd/libfmhr_ham.so
fi
timeout -k 10 ${TEST_TIMEOUT:-600} python -m pytest $R/tests ${PYTEST_X--x} -q -m gpu ${PYTEST_ARGS:-} > $R/gpurun_out/t.log 2>&1 || { tail -40 $R/gpurun_out/t.log; exit 1; }
tail -1 $R/gpurun_out/t.log
timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/b20.log 2>&1 || { tail -30 $R/gpurun_out/b20.log; exit 1; }
timeout -k 10 300 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/b200.log 2>&1 || { tail -30 $R/gpurun_out/b200.log; exit 1; }
python - <<PY
import json
for f in ("b20", "b200"):
    d = json.loads(open("$R/gpurun_out/%s.log" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 4), d["config"]["preroll_steps"], d["config"]["aa_items_per_step"], d["config"]["V"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
    print({k: round(v, 4) for k, v in d["kernels_ms"].items() if v > 0.001})
PY
