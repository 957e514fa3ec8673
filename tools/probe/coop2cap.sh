#!/bin/bash
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for n in 1280 1536 1280 1152 1408 2048; do
  echo "chunks=$n" | tee -a "$OUT/c.txt"
  timeout -k 10 200 python tools/prof_encode.py $n 1048576 2>&1 | grep positions | tee -a "$OUT/c.txt" || exit 1
done
