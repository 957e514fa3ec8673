#!/bin/bash
# the per-rank shares of the 2- and 4-GPU runs of the benchmark: 2048 and 1024 chunks of 1 MiB
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for n in 2048 1024; do
  timeout -k 10 200 python tools/prof_encode.py $n 1048576 2>&1 | grep positions | tee -a "$OUT/n2.txt" || exit 1
done
