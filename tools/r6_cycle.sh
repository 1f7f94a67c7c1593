#!/bin/bash
# Round-6 GPU cycle: the full GPU suite, the default bench, the demo loop bench, and the N > 1 rehearsals (ranks sharing the one GPU over gloo).
#   gpurun --timeout 1100 -- 'bash tools/r6_cycle.sh'       (TESTS="-k expr" restricts the suite; SKIP_TESTS=1 / SKIP_REHEARSE=1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -z "$SKIP_TESTS" ]; then
  rm -f $O/parity_r06.txt
  timeout -k 10 ${TEST_TIMEOUT:-700} python -m pytest ${TESTS:-$R/tests} -x -q -m gpu > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
  tail -1 $O/t.log
fi
if [ -z "$SKIP_SMOKE" ]; then
  timeout -k 10 200 python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
  tail -1 $O/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
  timeout -k 10 300 python $R/bench.py ${BENCH_ARGS:-} > $O/b.log 2>&1 || { tail -30 $O/b.log; exit 1; }
  timeout -k 10 300 python $R/bench.py --config demo --steps 60 --warmup 10 --no-stage1 ${DEMO_ARGS:-} > $O/b_demo.log 2>&1 || { tail -30 $O/b_demo.log; exit 1; }
fi
if [ -z "$SKIP_REHEARSE" ]; then
  FMHR_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage1 --no-run-length > $O/n1_rccl.log 2>&1 || { tail -30 $O/n1_rccl.log; exit 1; }
  for N in 2 4; do
    FMHR_DIST_BACKEND=gloo FMHR_FORCE_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29680 + N)) \
        $R/bench.py --gpus $N --steps 20 --warmup 5 > $O/n${N}_gloo.log 2>&1 || { tail -40 $O/n${N}_gloo.log; exit 1; }
    FMHR_DIST_BACKEND=gloo FMHR_FORCE_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29690 + N)) \
        $R/bench.py --config demo --gpus $N --steps 20 --warmup 5 > $O/n${N}_demo_gloo.log 2>&1 || { tail -40 $O/n${N}_demo_gloo.log; exit 1; }
  done
fi
python - <<PY
import json, os
for f in ("b", "b_demo", "n1_rccl", "n2_gloo", "n2_demo_gloo", "n4_gloo", "n4_demo_gloo"):
    path = "$O/%s.log" % f
    if not os.path.exists(path):
        continue
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no JSON line:", e); continue
    c = d.get("collective") or {}
    print(f, d["n_gpus"], round(d["value"], 1), round(d["ms_per_step"], 4), d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), c.get("path"), c.get("exchange_ok"), c.get("per_rank_ms"))
PY
