#!/bin/bash
# A/B/C... of prebuilt libraries (build/ab/<name>.so) on ONE box, REPS alternating rounds, then mean / min / max per variant of
# the step rate and of the launch groups in KEYS — k_face_bwd moves 75 -> 110 us from process to process, two rounds say nothing.
#   gpurun --timeout 1200 -- 'VARIANTS="base new" REPS=6 KEYS=shade_bwd bash tools/abstat.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/abstat
rm -f $R/gpurun_out/abstat/*.jsonl
for rep in $(seq 1 ${REPS:-6}); do for v in ${VARIANTS:-old new}; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  python $R/bench.py --no-cpu-baseline --no-stage1 --no-run-length ${BENCH_ARGS:-} > $R/gpurun_out/abstat/_run.log 2>&1 || { tail -5 $R/gpurun_out/abstat/_run.log; exit 1; }
  tail -1 $R/gpurun_out/abstat/_run.log >> $R/gpurun_out/abstat/$v.jsonl
  echo "round $rep $v done"
done; done
python - <<PY
import json, glob, os
keys = "${KEYS:-shade_bwd}".split(",")
for v in "${VARIANTS:-old new}".split():
    rows = [json.loads(l) for l in open("$R/gpurun_out/abstat/%s.jsonl" % v)]
    def st(x): return "%.1f [%.1f .. %.1f]" % (sum(x) / len(x), min(x), max(x))
    print(v.ljust(14), "it/s", st([r["value"] for r in rows]), " ".join("%s us %s" % (k, st([1e3 * r["kernels_ms"][k] for r in rows])) for k in keys))
PY
