#!/bin/bash
# same-box A/B of prebuilt libraries (build/ab/<name>.so) on the fused stage-1 step: VARIANTS="a b" bash tools/ab_stage1.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for v in ${VARIANTS}; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  echo "$v $(python $R/tools/stage1_rate.py ${VIEWS:-48} fusedonly 2>/dev/null | tail -1)"
done; done
