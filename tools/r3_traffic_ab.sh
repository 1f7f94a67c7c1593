#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the step with the atomic-key rasterizer and with the binned one, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for m in 0 1; do
  export FMHR_RASTER_BINNED=$m
  bash $R/tools/pmc_traffic.sh r03_binned$m > $R/gpurun_out/traffic_binned$m.log 2>&1 || { tail -20 $R/gpurun_out/traffic_binned$m.log; exit 1; }
  grep -E "k_raster_tris|k_raster_resolve|k_raster_bins|per-step|measured" $R/gpurun_out/r03_binned${m}_hbm_traffic.txt
done
