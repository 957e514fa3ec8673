#!/bin/bash
# Runs on the GPU box from the repository root: kernel trace and PMC passes of the
# match-finder benchmark into gpurun_out/<tag>/.   bash tools/prof_mf.sh <tag> [extra bench args]
# One rocprofv3 run per counter group; counters never together with a trace.
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
shift
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
ARGS="--workload mf --steps 1 --warmup 1 --no-cpu-baseline --no-verify $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o mf -- \
    python3 "$R/bench.py" $ARGS > "$OUT/ks_log.txt" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt"
for g in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
         "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
         "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${g%%:*}
    c=${g#*:}
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$n" -o p -- \
        python3 "$R/bench.py" $ARGS > "$OUT/pmc_$n.log" 2>&1
    cp "$(find "$OUT/pmc_$n" -name '*counter_collection.csv' | head -1)" "$OUT/$n.csv"
    rm -rf "$OUT/pmc_$n"
    echo "pmc pass $n done"
done
