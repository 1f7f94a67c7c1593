#!/bin/bash
# Is k_face_bwd's process-to-process spread the power-of-two view stride of the pixel arrays (512 x 512 x 16 B = exactly 4 MiB)?
# Alternating processes of the metric workload and of the same workload with 520 rows (4.06 MiB apart); HIP-event time per launch group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/stride
rm -f $R/gpurun_out/stride/*.jsonl
for rep in $(seq 1 ${REPS:-5}); do for c in metric metric_h520; do
  python $R/bench.py --config $c --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/stride/_run.log 2>&1 || { tail -5 $R/gpurun_out/stride/_run.log; exit 1; }
  tail -1 $R/gpurun_out/stride/_run.log >> $R/gpurun_out/stride/$c.jsonl
  echo "round $rep $c done"
done; done
python - <<PY
import json
for c in ("metric", "metric_h520"):
    rows = [json.loads(l) for l in open("$R/gpurun_out/stride/%s.jsonl" % c)]
    def st(x): return "%.1f [%.1f .. %.1f]" % (sum(x) / len(x), min(x), max(x))
    print(c.ljust(12), "it/s", st([r["value"] for r in rows]), " ".join("%s %s" % (k, st([1e3 * r["kernels_ms"][k] for r in rows])) for k in ("shade_bwd", "raster_tris", "raster_resolve", "loss_pix", "aa_fwd")))
PY
