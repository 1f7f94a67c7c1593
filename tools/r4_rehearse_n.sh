#!/bin/bash
# Rehearsal of the driver's N > 1 launch line on a one-GPU box: (1) RCCL with a world of one (FMHR_FORCE_SHARDED=1: every line of the
# sharded bench path incl. the negotiated direct communicator), (2) the exact torch.distributed.run line with 2 and 4 ranks sharing the
# one GPU over gloo (RCCL refuses two ranks on one device).   gpurun --timeout 900 -- 'bash tools/r4_rehearse_n.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
FMHR_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 timeout -k 10 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage1 --no-run-length > $O/n1_rccl.log 2>&1 || { tail -30 $O/n1_rccl.log; exit 1; }
for N in 2 4; do
  FMHR_DIST_BACKEND=gloo FMHR_FORCE_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29680 + N)) \
      $R/bench.py --gpus $N --steps 20 --warmup 5 > $O/n${N}_gloo.log 2>&1 || { tail -40 $O/n${N}_gloo.log; exit 1; }
done
python - <<PY
import json
for f in ("n1_rccl", "n2_gloo", "n4_gloo"):
    d = json.loads(open("$O/%s.log" % f).read().strip().splitlines()[-1])
    c = d.get("collective", {})
    print(f, d["n_gpus"], round(d["value"], 1), round(d["ms_per_step"], 4), d["scaling"], c.get("path"), c.get("path_negotiation"), c.get("per_rank_ms"))
PY
