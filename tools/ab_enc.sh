#!/bin/bash
# Runs on the GPU box: lone 1 MiB chain and 3072 chains of 256 KiB (one wave per chunk) for each
# library given, two rounds, for A/B comparisons on one box.   bash tools/ab_enc.sh <tag> lib...
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for round in 1 2; do
for lib in "$@"; do
  echo "== $lib" | tee -a "$OUT/ab.txt"
  LZMA_AMD_LIB=$lib LZMA_AMD_COOP=0 timeout -k 10 120 python tools/prof_encode.py 1 1048576 2>&1 | grep positions | tee -a "$OUT/ab.txt" || exit 1
  LZMA_AMD_LIB=$lib LZMA_AMD_COOP=0 timeout -k 10 200 python tools/prof_encode.py 3072 262144 2>&1 | grep positions | tee -a "$OUT/ab.txt" || exit 1
done
done
