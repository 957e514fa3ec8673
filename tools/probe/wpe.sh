#!/bin/bash
# A/B of register caps: every library given runs a lone chain, 3072 and 4096 chunks of 256 KiB at pb 0
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for lib in "$@"; do
  echo "== $lib" | tee -a "$OUT/wpe.txt"
  for shape in "1 1048576" "3072 262144" "4096 262144"; do
    LZMA_AMD_LIB=$lib LZMA_AMD_COOP=0 timeout -k 10 200 python tools/prof_encode.py $shape 22 128 markov - 0 2>&1 | grep positions | tee -a "$OUT/wpe.txt" || exit 1
  done
done
