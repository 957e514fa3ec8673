#!/bin/bash
# Runs on the GPU box from the repository root: raw profiling outputs of the encode and
# match-finder and decode benchmarks into gpurun_out/<tag>/ (condensed afterwards by tools/make_profiles.py).
#   bash tools/collect_profiles.sh <tag> [round]
# With a round name (r02) the summaries are made on the box as well, into gpurun_out/<tag>/profiles/, and the
# large raw traces are dropped (gpurun_out/ carries at most 64 MiB back).
# One rocprofv3 run per counter group; counters never together with a trace.
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
mkdir -p "$OUT/pmc"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 python3 "$R/bench.py" > "$OUT/bench_default.txt" 2> "$OUT/bench_default.err"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o enc -- \
    python3 "$R/bench.py" --no-cpu-baseline --no-verify --no-extra > "$OUT/ks_enc_log.txt" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_raw.csv"
cp "$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)" "$OUT/kernel_trace.csv"
rm -rf "$OUT/kt"
timeout -k 10 300 python3 "$R/bench.py" --workload mf > "$OUT/bench_mf.txt" 2> "$OUT/bench_mf.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktm" -o mf -- \
    python3 "$R/bench.py" --workload mf --no-cpu-baseline --no-verify > "$OUT/ks_mf_log.txt" 2>&1
cp "$(find "$OUT/ktm" -name '*kernel_stats.csv' | head -1)" "$OUT/mf_kernel_stats_raw.csv"
rm -rf "$OUT/ktm"
for g in "fetch:FETCH_SIZE" "write:WRITE_SIZE" \
         "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
         "l2:TCC_HIT_sum TCC_MISS_sum"; do
    n=${g%%:*}
    c=${g#*:}
    # (one wave per chunk, as in the full benchmark: 832 chunks alone would get two)
    LZMA_AMD_COOP=0 timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$n" -o p -- \
        python3 "$R/bench.py" --chunks 832 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra --no-e2e > "$OUT/pmc_$n.log" 2>&1
    cp "$(find "$OUT/pmc_$n" -name '*counter_collection.csv' | head -1)" "$OUT/pmc/$n.csv"
    rm -rf "$OUT/pmc_$n"
    echo "pmc pass $n done"
done
timeout -k 10 300 python3 "$R/bench.py" --workload decode > "$OUT/bench_decode.txt" 2> "$OUT/bench_decode.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktd" -o dec -- \
    python3 "$R/bench.py" --workload decode --no-cpu-baseline --no-verify > "$OUT/ks_dec_log.txt" 2>&1
cp "$(find "$OUT/ktd" -name '*kernel_stats.csv' | head -1)" "$OUT/dec_kernel_stats_raw.csv"
rm -rf "$OUT/ktd"
mkdir -p "$OUT/pmc_dec"
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
mkdir -p "$OUT/pmc_mf"
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
if [ -n "$2" ]; then
    python3 "$R/tools/make_profiles.py" "$OUT" "$2" 832 "$OUT/profiles" > "$OUT/make_profiles.txt" 2>&1
    rm -f "$OUT/kernel_trace.csv" "$OUT"/pmc/*.csv "$OUT"/pmc_mf/*.csv
fi
