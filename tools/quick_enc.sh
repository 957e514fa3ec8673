#!/bin/bash
# Runs on the GPU box from the repository root: times of a lone 1 MiB chain (one wave, two
# waves), of 1024 chains of 1 MiB with two waves each and of 3072 chains of 256 KiB (every
# resident slot taken), then the encoder parity tests with every launch forced to two waves
# per chunk and as the library chooses.
#   bash tools/quick_enc.sh <tag> [lib]
set -o pipefail
OUT=gpurun_out/$1
mkdir -p "$OUT"
[ -n "$2" ] && export LZMA_AMD_LIB=$2
LZMA_AMD_COOP=0 timeout -k 10 120 python tools/prof_encode.py 1 1048576 > "$OUT/lone.txt" 2>&1 || { tail -5 "$OUT/lone.txt"; exit 1; }
LZMA_AMD_COOP=2 timeout -k 10 120 python tools/prof_encode.py 1 1048576 22 128 markov verify > "$OUT/lone_coop.txt" 2>&1 || { tail -5 "$OUT/lone_coop.txt"; exit 1; }
LZMA_AMD_COOP=3 timeout -k 10 120 python tools/prof_encode.py 1 1048576 22 128 markov verify > "$OUT/lone_coop3.txt" 2>&1 || { tail -5 "$OUT/lone_coop3.txt"; exit 1; }
LZMA_AMD_COOP=3 timeout -k 10 200 python tools/prof_encode.py 512 1048576 > "$OUT/coop3_512.txt" 2>&1 || { tail -5 "$OUT/coop3_512.txt"; exit 1; }
LZMA_AMD_COOP=2 timeout -k 10 200 python tools/prof_encode.py 1024 1048576 > "$OUT/coop1024.txt" 2>&1 || { tail -5 "$OUT/coop1024.txt"; exit 1; }
LZMA_AMD_COOP=0 timeout -k 10 200 python tools/prof_encode.py 3072 262144 > "$OUT/full.txt" 2>&1 || { tail -5 "$OUT/full.txt"; exit 1; }
grep -h "positions\|round trip" "$OUT/lone.txt" "$OUT/lone_coop.txt" "$OUT/lone_coop3.txt" "$OUT/coop3_512.txt" "$OUT/coop1024.txt" "$OUT/full.txt"
timeout -k 10 600 python -u -m pytest tests/test_gpu_encode.py -x -q --timeout 300 --timeout-method thread > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
