#!/bin/bash
# frames of a sequence capture on several HIP streams: does the chip have slack that independent problems can fill?
set -e
mkdir -p gpurun_out/streams
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stage1 --no-run-length"
for k in 1 2; do FMHR_DUP_FRAMES=2 $B --frame-streams $k > gpurun_out/streams/metric_dup2_s$k.json; done
for k in 1 2 4 8; do $B --config C5 --frame-streams $k > gpurun_out/streams/C5_s$k.json; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/streams/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"])
PY
