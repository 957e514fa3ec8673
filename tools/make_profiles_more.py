"""Condenses what tools/collect_profiles_more.sh left in gpurun_out/<tag>/ into profiles/:
    python tools/make_profiles_more.py gpurun_out/<tag> <round> [profiles]
"""
import ast
import json
import shutil
import sys
from pathlib import Path

run, tag = Path(sys.argv[1]), sys.argv[2]
prof = Path(sys.argv[3] if len(sys.argv) > 3 else "profiles")


def bench_line(path):
    return [ln for ln in path.read_text(errors="replace").splitlines() if ln.startswith('{"metric"')][-1]


line = bench_line(run / "bench_coop512.txt")
txt = (run / "coop512_pmc_sq_per_input_byte.txt").read_text()
raw = ast.literal_eval(txt[txt.index("{"):])
pmc = {k: round(v * (4 if "CYCLES" in k or k == "SQ_WAIT_ANY" else 1), 1) for k, v in raw.items()}
one = json.load(open(prof / f"{tag}_pmc_encode_832x1MiB.json"))["kernels"]["encode_kernel"]["per_position"]
res = {
    "command": "python bench.py --chunks 512 --steps 2 --warmup 1 --no-extra --no-cpu-baseline   (counters: one SQ pass, "
               "rocprofv3 --pmc ..., on --chunks 512 --steps 1 --warmup 0)",
    "workload": "encode, 512 chunks of 1 MiB (one GPU's share of the 4 GiB benchmark on eight GPUs): the library picks "
                "three waves per chunk",
    "bench_line": json.loads(line),
    "pmc_sq_per_input_byte_all_waves": pmc,
    "notes": [
        "counter values are per input byte, summed over the three waves of a workgroup; cycle counters multiplied out to "
        "cycles (rocprofv3 reports units of four)",
        f"one wave per chunk on the same text: {one['valu']} VALU + {one['salu']} SALU + {one['lds']} LDS instructions and "
        f"{one['wave_cycles']} wave cycles per byte ({tag}_pmc_encode_832x1MiB.json)",
    ],
}
(prof / f"{tag}_encode_coop3_512x1MiB.json").write_text(json.dumps(res, indent=1) + "\n")
shutil.copy(run / "coop512_kernel_stats.csv", prof / f"{tag}_encode_coop3_512x1MiB_kernel_stats.csv")
if (run / "bench_decode_1M.txt").exists():
    res = {"command": "python bench.py --workload decode --streams 1000000 --steps 2 --warmup 1",
           "bench_line": json.loads(bench_line(run / "bench_decode_1M.txt"))}
    (prof / f"{tag}_bench_decode_1M_streams.json").write_text(json.dumps(res, indent=1) + "\n")
print("ok")


def c5(run5):
    """... and what tools/collect_c5.sh left (a directory given as the fourth argument)."""
    import csv
    import re
    rows = []
    for r in csv.DictReader(open(run5 / "c5_kernel_stats_raw.csv")):
        m = re.search(r"(encode_kernel<[^>]*>|(mf|rs)_\w+_kernel|radix_sort_onesweep\w*|scan_impl|__amd_rocclr_\w+)", r["Name"])
        rows.append({"kernel": m.group(1) if m else r["Name"][:60], "calls": int(r["Calls"]),
                     "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 1), "avg_ms": round(float(r["AverageNs"]) / 1e6, 2)})
    pmc = {}
    for name, f in (("encode", "c5_pmc_sq_encode.txt"), ("scan", "c5_pmc_sq_scan.txt")):
        t = (run5 / f).read_text()
        if "{" in t:
            pmc[name] = ast.literal_eval(t[t.index("{"):t.index("}") + 1])
    res = {
        "command": "python bench.py --workload c5 --chunks 256 --steps 1 --warmup 0 --no-extra --no-e2e   (kernel stats: the same "
                   "under rocprofv3 --kernel-trace --stats with --no-cpu-baseline --no-verify; counters: one SQ pass on --chunks 64)",
        "workload": "configs[4]: 256 chunks of 16 MiB redundant dpkg-style log = 4 GiB, dict 64 MiB, fb 273, one MI355X",
        "bench_line": json.loads(bench_line(run5 / "bench_c5.txt")),
        "kernel_stats": rows[:16],
        "pmc_sq_64_chunks": pmc,
        "notes": ["SQ cycle counters in units of four cycles, summed over all waves of the kernel",
                  "a chunk is one dependent chain: three waves work on it (encode_kernel<.., 3u>), batches are launched as their "
                  "match lists are ready"],
    }
    (prof / f"{tag}_c5_256x16MiB.json").write_text(json.dumps(res, indent=1) + "\n")


if len(sys.argv) > 4:
    c5(Path(sys.argv[4]))
    print("c5 ok")
