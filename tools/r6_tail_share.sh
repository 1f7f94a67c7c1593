#!/bin/bash
# VERDICT r5 item 2, second half: what the regularisers cost INSIDE the mesh tails of a view-sharded rank — the 6-view share of the metric
# run on ONE rank through the split backward (FMHR_FORCE_SHARDED=1: RCCL world of one, combine kernel), with the regularisers of
#   all vertices (K = 1) | an eighth of them (K = 8: today's split by vertex range at 8 GPUs) | none (K = 0: what a form that moves them to
#   a second stream would leave on the main stream).   gpurun --timeout 900 -- 'bash tools/r6_tail_share.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
for K in 1 8 0; do
  ENVS="FMHR_FORCE_SHARDED=1 FMHR_BENCH_REG_SHARE=$K MASTER_ADDR=127.0.0.1 MASTER_PORT=2968$K" TAG=r06_tail_share_K$K STEPS=200 BENCH_ARGS="--views 6 --no-run-ham" bash $R/tools/kstats.sh > $R/gpurun_out/r06_tail_share_K$K.out 2>&1 || { tail -20 $R/gpurun_out/r06_tail_share_K$K.out; exit 1; }
done
python3 - <<PY | tee $R/gpurun_out/r06_tail_share.txt
import csv
print("# 6 views x 512 x 512 of the metric mesh on one rank, split backward (RCCL world of one): mean us per launch over 200 timed steps (rocprofv3 kernel trace)")
print("# K = regulariser share of the rank: 1 all vertices | 8 an eighth (8-GPU split by vertex range) | 0 none (regularisers elsewhere)")
rows = {}
for K in (1, 8, 0):
    for r in csv.DictReader(open("$R/gpurun_out/r06_tail_share_K%d_kernel_stats_timed.csv" % K)):
        rows.setdefault(r["Name"], {})[K] = float(r["AverageNs"]) / 1e3
print("kernel".ljust(40), "K=1".rjust(8), "K=8".rjust(8), "K=0".rjust(8))
for name, d in rows.items():
    print(name[:40].ljust(40), *("%8.1f" % d.get(K, float("nan")) for K in (1, 8, 0)))
PY
