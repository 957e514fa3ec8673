"""Turns the raw outputs of one profiling run into the summaries kept under profiles/.

    python tools/make_profiles.py gpurun_out/<run> r01 832

The run directory holds what the profiling command writes on the GPU box:
  bench_default.txt / bench_mf.txt / bench_decode.txt   stdout of the three bench.py runs
  kernel_stats.csv, timeline.txt                        from rocprofv3 --kernel-trace --stats on bench.py
  pmc/{fetch,write,sq,l2}.csv                           rocprofv3 --pmc passes on bench.py --chunks N --steps 1
"""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def kernel_class(name):
    if "encode_kernel" in name:
        return "encode_kernel"
    if "mf_scan" in name:
        return "mf_scan_kernel"
    if "rocprim" in name:
        return "rocprim_sorts_and_scans"
    if "mf_" in name:
        return "mf_table_kernels"
    return "other"


def main():
    run, tag, chunks = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    for wl in ("default", "mf", "decode"):
        lines = [l for l in (run / f"bench_{wl}.txt").read_text(errors="replace").splitlines() if l.startswith('{"metric"')]
        (prof / f"{tag}_bench_{'encode' if wl == 'default' else wl}.json").write_text(lines[-1] + "\n")
    (prof / f"{tag}_encode_default_kernel_stats.csv").write_text(
        "# rocprofv3 --kernel-trace --stats -- python3 bench.py   (defaults: 4096 x 1 MiB chunks, 1 warmup + 2 timed "
        "steps; MI355X; durations in microseconds; view top_kernels of the rocpd database)\n"
        + (run / "kernel_stats.csv").read_text())
    (prof / f"{tag}_encode_default_timeline.txt").write_text(
        "# kernel timeline of the same run (ms since the first kernel): E = encode_kernel launch (start end duration "
        "grid), scan = mf_scan_kernel, k = other finder kernels over 30 ms\n" + (run / "timeline.txt").read_text())
    out = {}
    for d in ("fetch", "write", "sq", "l2"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls, seen = collections.Counter(), set()
        for row in csv.DictReader(open(run / "pmc" / f"{d}.csv")):
            k = kernel_class(row["Kernel_Name"])
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if (k, row["Dispatch_Id"]) not in seen:
                seen.add((k, row["Dispatch_Id"]))
                calls[k] += 1
        for k in agg:
            out.setdefault(k, {}).update(agg[k])
            out[k]["launches"] = calls[k]
    n_in = chunks << 20
    rows = {}
    for k in ("encode_kernel", "mf_scan_kernel", "rocprim_sorts_and_scans", "mf_table_kernels"):
        v = out[k]
        rows[k] = {
            "launches": v["launches"], "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
            "hbm_bytes_per_input_byte_reported": round((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n_in, 2),
            "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4),
            "TCC_HIT_sum": v["TCC_HIT_sum"], "TCC_MISS_sum": v["TCC_MISS_sum"],
            "SQ_WAVE_CYCLES": v["SQ_WAVE_CYCLES"], "SQ_BUSY_CYCLES": v["SQ_BUSY_CYCLES"], "SQ_WAIT_ANY": v["SQ_WAIT_ANY"],
            "SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "SQ_INSTS_SALU": v["SQ_INSTS_SALU"], "SQ_INSTS_LDS": v["SQ_INSTS_LDS"],
            "SQ_LDS_IDX_ACTIVE": v["SQ_LDS_IDX_ACTIVE"], "SQ_LDS_BANK_CONFLICT": v["SQ_LDS_BANK_CONFLICT"],
            "per_position": {"valu": round(v["SQ_INSTS_VALU"] / n_in, 1), "salu": round(v["SQ_INSTS_SALU"] / n_in, 1),
                             "lds": round(v["SQ_INSTS_LDS"] / n_in, 2),
                             "wave_cycles": round(4 * v["SQ_WAVE_CYCLES"] / n_in),
                             "wait_cycles": round(4 * v["SQ_WAIT_ANY"] / n_in)}}
    res = {
        "command": f"rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --chunks {chunks} --steps 1 "
                   "--warmup 0 --no-cpu-baseline --no-verify  (one run per counter group: FETCH_SIZE | WRITE_SIZE | "
                   "SQ_* | TCC_HIT_sum TCC_MISS_sum)",
        "workload": f"encode, {chunks} chunks of 1 MiB (one finder batch, one encode launch), MI355X",
        "input_bytes": n_in,
        "notes": ["FETCH_SIZE/WRITE_SIZE are in KB as rocprofv3 reports them, summed over the kernel's dispatches",
                  "FETCH_SIZE on gfx950 reports half of the bytes of 16-byte-per-lane streaming reads; the reads here "
                  "are 12-byte entries and narrower, which are uncalibrated: treat the figures as a lower bound up to "
                  "a factor of two",
                  "SQ_* cycle counters are in units of four cycles, summed over all waves",
                  "LDS traffic of encode_kernel: SQ_INSTS_LDS wave instructions, most of them with one or a few "
                  "active lanes"],
        "kernels": rows}
    (prof / f"{tag}_pmc_encode_{chunks}x1MiB.json").write_text(json.dumps(res, indent=1) + "\n")
    for k, v in rows.items():
        print(k, v["launches"], "B/byte", v["hbm_bytes_per_input_byte_reported"], "L2", v["l2_hit_rate"],
              v["per_position"], "fetchGB", round(v["FETCH_SIZE_KB"] * 1024 / 1e9, 1), "writeGB",
              round(v["WRITE_SIZE_KB"] * 1024 / 1e9, 1))


if __name__ == "__main__":
    main()
