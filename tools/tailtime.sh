R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-v66 tail2}; do
  cp $R/build/ab/$v.so $R/fmhr_amd/libfmhr_ham.so
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tt_$v -o tt --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stage1 --no-run-length > $R/gpurun_out/tt_$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/tt_$v/**/*kernel_stats.csv", recursive=True)[0]
r = {x["Name"]: float(x["AverageNs"]) / 1000 for x in csv.DictReader(open(f))}
print("$v".ljust(12), {k.split("(")[0].replace("fmhr::","")[:16]: round(v, 1) for k, v in r.items() if "mesh_tail" in k or "k_geometry" in k or "k_adam" in k})
PY
done
