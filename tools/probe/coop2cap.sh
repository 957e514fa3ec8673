#!/bin/bash
# what the library chooses from 512 to 1536 chunks of 1 MiB
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for n in 512 1024 1280 1536; do
  echo "chunks=$n" | tee -a "$OUT/c.txt"
  timeout -k 10 200 python tools/prof_encode.py $n 1048576 2>&1 | grep positions | tee -a "$OUT/c.txt" || exit 1
done
