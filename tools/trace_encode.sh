#!/bin/bash
# Runs on the GPU box from the repository root: kernel trace of one encode benchmark step
# into gpurun_out/<tag>/kernel_trace.csv.   bash tools/trace_encode.sh <tag> [extra bench args]
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o enc -- \
    python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-extra $* > "$OUT/log.txt" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)" "$OUT/kernel_trace.csv"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt"
