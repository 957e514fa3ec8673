#!/bin/bash
# One rocprofv3 counter pass over a command, summed per kernel:
#   bash tools/pmc_pass.sh <tag> "<counters>" <kernel substring> <divisor> -- python3 <script> [args]
set -e -o pipefail
R=$(pwd)
TAG=$1; CNT=$2; SUB=$3; DIV=$4; shift 5
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/raw" -o p -- "$@" > "$OUT/log.txt" 2>&1
python3 "$R/tools/pmc_sum.py" "$OUT/raw" "$SUB" "$DIV" | tee "$OUT/sum.txt"
rm -rf "$OUT/raw"
