#!/bin/bash
# Samples the GPU's engine clock while an encode call runs:  bash tools/clock_sample.sh <nchunks>
# (reads only; prints rocm-smi's sclk line every two seconds)
R=$(pwd)
LZMA_AMD_CO_RUN=0 timeout -k 10 120 python3 "$R/tools/prof_encode.py" "$1" 1048576 > /tmp/clk_run.txt 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4; do
    /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1
    sleep 2
done
wait $PID
grep positions /tmp/clk_run.txt
