#!/bin/bash
# Runs on the GPU box from the repository root: the redundant-log workload (16 MiB chunks,
# dictionary 64 MiB, 273 fast bytes) at 256 chunks = 4 GiB: bench line with roofline and CPU
# baseline, kernel statistics of the same command, one counter pass on 64 chunks.
#   bash tools/collect_c5.sh <tag>
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 700 python3 "$R/bench.py" --workload c5 --chunks 256 --steps 1 --warmup 0 --no-extra --no-e2e > "$OUT/bench_c5.txt" 2> "$OUT/bench_c5.err"
tail -c 600 "$OUT/bench_c5.txt"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o c5 -- \
    python3 "$R/bench.py" --workload c5 --chunks 256 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra --no-e2e > "$OUT/ks_c5_log.txt" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/c5_kernel_stats_raw.csv"
rm -rf "$OUT/kt"
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/pmc_sq" -o p -- \
    python3 "$R/bench.py" --workload c5 --chunks 64 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra --no-e2e > "$OUT/pmc_sq.log" 2>&1
python3 "$R/tools/pmc_sum.py" "$OUT/pmc_sq" encode_kernel > "$OUT/c5_pmc_sq_encode.txt" || true
python3 "$R/tools/pmc_sum.py" "$OUT/pmc_sq" mf_scan_kernel > "$OUT/c5_pmc_sq_scan.txt" || true
rm -rf "$OUT/pmc_sq"
echo done
