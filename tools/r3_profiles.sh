#!/bin/bash
# Round-3 profile set for the committed library: rocprofv3 kernel stats of the bench command, the two HBM-traffic passes, the
# issue-counter passes.  TAG=r03_v82 tools/r3_profiles.sh   (outputs under gpurun_out/, copied into profiles/ by hand)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r03_vXX}
mkdir -p $R/gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_p.log ) || { tail -20 $R/gpurun_out/${TAG}_p.log; exit 1; }
cp $(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -14 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
bash $R/tools/pmc_traffic.sh $TAG | tail -14 && python3 $R/tools/pmc_issue.py $TAG | tail -5
