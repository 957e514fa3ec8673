#!/bin/bash
# The four counter passes of tools/collect_profiles.sh on the match-finder benchmark alone:
#   bash tools/collect_pmc_mf.sh <tag>      (raw CSVs into gpurun_out/<tag>/pmc_mf/)
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
mkdir -p "$OUT/pmc_mf"
cd /tmp
export TMPDIR=/tmp
for g in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
         "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${g%%:*}
    c=${g#*:}
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmcm_$n" -o p -- \
        python3 "$R/bench.py" --workload mf --steps 1 --warmup 0 --no-cpu-baseline --no-verify > "$OUT/pmcm_$n.log" 2>&1
    cp "$(find "$OUT/pmcm_$n" -name '*counter_collection.csv' | head -1)" "$OUT/pmc_mf/$n.csv"
    rm -rf "$OUT/pmcm_$n"
    echo "mf pmc pass $n done"
done
