#!/bin/bash
# Row pitch (ABI 5) vs the flat ABI-4 layout on the W = 334 configurations, same box, two alternating rounds each:
#   gpurun --timeout 900 -- 'bash tools/r4_pitch_ab.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pitch_ab
KEYS="geometry raster_tris raster_large raster_resolve shade_fwd aa_fwd loss_pix aa_bwd shade_bwd regularisers"
for cfg in ${CONFIGS:-C2 C5}; do
  for rep in 1 2; do for v in 0 1; do
    FMHR_ROW_PITCH=$v python $R/bench.py --config $cfg --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/pitch_ab/${cfg}_pitch${v}_$rep.log 2>&1 || { tail -5 $R/gpurun_out/pitch_ab/${cfg}_pitch${v}_$rep.log; exit 1; }
    tail -1 $R/gpurun_out/pitch_ab/${cfg}_pitch${v}_$rep.log > $R/gpurun_out/pitch_ab/${cfg}_pitch${v}_$rep.json
    python - <<PY
import json
d = json.load(open("$R/gpurun_out/pitch_ab/${cfg}_pitch${v}_$rep.json"))
k = d["kernels_ms"]
print("$cfg", "pitch" if $v else "flat ", round(d["value"], 1), "it/s", round(d["ms_per_step"], 4), "ms |", d["config"]["rasterizer"], "|", " ".join("%s %.1f" % (x, 1e3 * k[x]) for x in "$KEYS".split()))
PY
  done; done
done
