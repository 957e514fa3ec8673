#!/bin/bash
# launch-order knobs on the 4096-chunk call
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for v in 1 0 2 3; do
  echo "CO_RUN=$v" | tee -a "$OUT/corun.txt"
  LZMA_AMD_CO_RUN=$v timeout -k 10 200 python tools/prof_encode.py 4096 1048576 2>&1 | grep positions | tee -a "$OUT/corun.txt" || exit 1
done
