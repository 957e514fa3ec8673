#!/bin/bash
# single-generation calls (2048 and 3072 chunks of 1 MiB): one launch beside the finder, or none
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for n in 2048 3072; do
for v in 1 0; do
  echo "chunks=$n CO_RUN=$v" | tee -a "$OUT/corun.txt"
  LZMA_AMD_CO_RUN=$v timeout -k 10 200 python tools/prof_encode.py $n 1048576 2>&1 | grep positions | tee -a "$OUT/corun.txt" || exit 1
done
done
