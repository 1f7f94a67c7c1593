set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
timeout -k 10 600 python -m pytest $R/tests/test_gpu_ham.py $R/tests/test_gpu_fuzz.py $R/tests/test_gpu_pipeline.py $R/tests/test_gpu_boundary.py -x -q -m gpu > $O/t_few.log 2>&1 || { tail -40 $O/t_few.log; exit 1; }
tail -1 $O/t_few.log
for cfg in "--views 6" "--config demo_batch --views 1" "--config C4 --views 6" "--config C2" "--config C1"; do
  for few in 0 1; do
    FMHR_FACE_BWD_FEW=$few python $R/bench.py $cfg --steps 100 --no-cpu-baseline --no-stage1 --no-run-length --no-run-ham > $O/few.log 2>&1 || { tail -5 $O/few.log; exit 1; }
    tail -1 $O/few.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$cfg few=$few'.ljust(44), round(d['value'],1), round(k['shade_bwd']*1e3,1), round(k['regularisers']*1e3,1))"
  done
done
