#!/bin/bash
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
timeout -k 10 300 python -u -m pytest tests/test_gpu_encode.py -x -q -k "more_waves or ragged" --timeout 200 --timeout-method thread > "$OUT/pytest.txt" 2>&1 || { tail -30 "$OUT/pytest.txt"; exit 1; }
tail -3 "$OUT/pytest.txt"
for n in 2048 2560; do
  timeout -k 10 200 python tools/prof_encode.py $n 1048576 22 128 markov verify 2>&1 | grep "positions\|round trip" | tee -a "$OUT/n2.txt" || exit 1
done
