#!/bin/bash
# Runs on the GPU box from the repository root: times of a lone 1 MiB chain and of 3072 chains of
# 256 KiB (every resident slot taken), then the encoder parity tests.
#   bash tools/quick_enc.sh <tag> [lib]
set -o pipefail
OUT=gpurun_out/$1
mkdir -p "$OUT"
[ -n "$2" ] && export LZMA_AMD_LIB=$2
timeout -k 10 120 python tools/prof_encode.py 1 1048576 > "$OUT/lone.txt" 2>&1 || { tail -5 "$OUT/lone.txt"; exit 1; }
timeout -k 10 200 python tools/prof_encode.py 3072 262144 > "$OUT/full.txt" 2>&1 || { tail -5 "$OUT/full.txt"; exit 1; }
grep -h positions "$OUT/lone.txt" "$OUT/full.txt"
timeout -k 10 600 python -u -m pytest tests/test_gpu_encode.py -x -q --timeout 300 --timeout-method thread > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
