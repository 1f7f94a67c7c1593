#!/bin/bash
# counter evidence for the per-face record experiment (tools/experiments/r06_face_record.patch) — L2 hit rate, wait cycles
# and duration of the second rasterizer pass at C4 with the tree's library ("base") and with the record build ("rec", build/ab/rec.so).
#   tools/build_variants.sh base= rec=-DFMHR_X_FACE_REC   (with the patch applied), then   gpurun --timeout 1000 -- 'bash tools/r6_face_record_counters.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in base rec; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  PASSES=0,1,3 BENCH_ARGS="--config C4 --no-run-ham" PASS_TIMEOUT=280 python3 $R/tools/pmc_issue.py r06_rec_$v > $R/gpurun_out/r06_rec_$v.out 2>&1 || { tail -5 $R/gpurun_out/r06_rec_$v.out; exit 1; }
  grep -E "^kernel|k_raster_bins|k_face_records|k_face_bwd" $R/gpurun_out/r06_rec_${v}_issue.txt | cut -c1-250
  rm -rf $R/gpurun_out/r06_rec_${v}_issue[0-9]
done
cp $R/build/ab/base.so $R/fmhr_amd/libfmhr_ham.so
