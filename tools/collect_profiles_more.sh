#!/bin/bash
# Runs on the GPU box from the repository root, after tools/collect_profiles.sh: the cooperative
# encode kernel on 512 chunks (one GPU's share of the benchmark on eight) and the whole
# small-stream decode batch, into gpurun_out/<tag>/.   bash tools/collect_profiles_more.sh <tag>
set -e -o pipefail
R=$(pwd)
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
# the cooperative kernel (three waves per chunk) on the share of one GPU of eight: bench line,
# kernel statistics, one counter pass
timeout -k 10 300 python3 "$R/bench.py" --chunks 512 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > "$OUT/bench_coop512.txt" 2> "$OUT/bench_coop512.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktc" -o c -- \
    python3 "$R/bench.py" --chunks 512 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-verify --no-e2e > "$OUT/ks_coop_log.txt" 2>&1
python3 - "$OUT" <<'PYEOF'
import csv, glob, re, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/ktc/**/*kernel_stats.csv", recursive=True)[0])))
with open(out + "/coop512_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_ms,percent\n")
    for r in rows[:12]:
        m = re.search(r"(encode_kernel<[^>]*>|decode_kernel<[^>]*>|(mf|rs)_\w+_kernel|radix_sort_onesweep\w*|scan_impl|__amd_rocclr_\w+)", r["Name"])
        f.write(f'{m.group(1) if m else r["Name"][:50]},{r["Calls"]},{float(r["TotalDurationNs"]) / 1e6:.2f},{float(r["AverageNs"]) / 1e6:.3f},{r["Percentage"]}\n')
PYEOF
rm -rf "$OUT/ktc"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/pmc_coop" -o p -- \
    python3 "$R/bench.py" --chunks 512 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra --no-e2e > "$OUT/pmc_coop.log" 2>&1
python3 "$R/tools/pmc_sum.py" "$OUT/pmc_coop" encode_kernel 536870912 > "$OUT/coop512_pmc_sq_per_input_byte.txt" || true
rm -rf "$OUT/pmc_coop"
# the whole small-stream batch of the decode configuration on one GPU
timeout -k 10 900 python3 "$R/bench.py" --workload decode --streams 1000000 --steps 2 --warmup 1 > "$OUT/bench_decode_1M.txt" 2> "$OUT/bench_decode_1M.err" || true
echo done
