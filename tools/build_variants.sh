#!/bin/bash
# Builds one library per set of extra compile flags into build/ab/<name>.so (for tools/abn.sh): name=flags pairs on the command line.
#   tools/build_variants.sh base= bb128=-DFMHR_BINS_BLOCK=128 "fb512=-DFMHR_FB_BLOCK=512"
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/ab
cp $R/fmhr_amd/libfmhr_ham.so $R/build/ab/.keep.so 2>/dev/null || true
for kv in "$@"; do
  name=${kv%%=*}; flags=${kv#*=}
  FMHR_EXTRA_FLAGS="$flags" FMHR_FORCE_BUILD=1 $R/fmhr_amd/csrc/build.sh > /dev/null 2>&1 || { echo "build failed: $name ($flags)"; continue; }
  cp $R/fmhr_amd/libfmhr_ham.so $R/build/ab/$name.so
  echo "built $name ($flags)"
done
FMHR_FORCE_BUILD=1 $R/fmhr_amd/csrc/build.sh > /dev/null 2>&1          # the tree's own library again
