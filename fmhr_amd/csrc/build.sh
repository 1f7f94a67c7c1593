#!/bin/bash
# fmhr_amd/csrc/build.sh — builds fmhr_amd/libfmhr_ham.so for gfx950 (no GPU needed: hipcc cross-compiles).
# -ffp-contract=off is part of the numerical contract: do not remove it.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libfmhr_ham.so
SRCS="abi.cpp raster.hip antialias.hip mesh.hip shade.hip ncc.hip ham_step.hip stage1.hip peer.hip"
NEW=0
for f in $SRCS common.h shade_common.h adam.h ../../include/fmhr_ham.h build.sh; do
  if [ ! -f "$OUT" ] || [ "$f" -nt "$OUT" ]; then NEW=1; fi
done
if [ "$NEW" = "0" ] && [ "${FMHR_FORCE_BUILD:-0}" = "0" ]; then echo "libfmhr_ham.so up to date"; exit 0; fi
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wall -Wno-unused-result ${FMHR_EXTRA_FLAGS:-}"
# raster.hip / antialias.hip hold every discrete decision (snap, coverage, depth order, antialias analysis): strict, one IEEE
# operation per statement.  The value-only kernels may fuse a*b+c WITHIN an expression (never across statements, so the shared
# decision helpers of common.h stay exact even there).
contract() { case "$1" in raster.hip|antialias.hip|abi.cpp) echo "-ffp-contract=off";; *) echo "-ffp-contract=on";; esac; }
mkdir -p ../../build/obj
OBJS=""
pids=""
for f in $SRCS; do
  o=../../build/obj/${f%.*}.o
  OBJS="$OBJS $o"
  ( $HIPCC $FLAGS $(contract $f) -x hip -c $f -o $o ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(realpath $OUT)"
