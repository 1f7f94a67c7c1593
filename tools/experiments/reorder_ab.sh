set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for r in 1 2 3; do
  for s in ${MODES:-0 1 2}; do
    FMHR_X_REORDER=$s timeout -k 10 200 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage1 --no-run-length > $O/reo_${s}_$r.log 2>&1 || { tail -20 $O/reo_${s}_$r.log; exit 1; }
  done
done
python - <<PY
import json, statistics
for s in [int(x) for x in "${MODES:-0 1 2}".split()]:
    ds = [json.loads(open("$O/reo_%d_%d.log" % (s, r)).read().strip().splitlines()[-1]) for r in range(1, 4)]
    print("reorder", s, [round(d["value"]) for d in ds], {k: round(v*1e3,1) for k, v in ds[-1]["kernels_ms"].items() if v > 0.006})
PY
