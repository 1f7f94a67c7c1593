R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "--views 6" "--config demo_batch --views 1"; do
BENCH_ARGS="$cfg --steps 100" VARIANTS="base nostore" KEYS=shade_bwd,regularisers bash $R/tools/abn.sh
done
