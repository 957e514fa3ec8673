#!/bin/bash
# Runs on the GPU box from the repository root: the four counter passes over one step of the
# decode benchmark (one decode_kernel launch), summed into gpurun_out/<tag>/pmc_dec/*.txt.
#   bash tools/collect_decode_pmc.sh <tag>
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
mkdir -p "$OUT/pmc_dec"
cd /tmp
export TMPDIR=/tmp
for g in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
         "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" \
         "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${g%%:*}
    c=${g#*:}
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmcd_$n" -o p -- \
        python3 "$R/bench.py" --workload decode --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-e2e > "$OUT/pmcd_$n.log" 2>&1
    python3 "$R/tools/pmc_sum.py" "$OUT/pmcd_$n" decode_kernel > "$OUT/pmc_dec/$n.txt"
    rm -rf "$OUT/pmcd_$n"
    echo "decode pmc pass $n done"
done
