"""Turns the raw outputs of tools/collect_profiles.sh into the encode and match-finder summaries kept under profiles/.

    python tools/make_profiles.py gpurun_out/<run> r01 832 [output directory, default profiles/]

The run directory holds what the collection script writes on the GPU box:
  bench_default.txt                       stdout of a plain bench.py run
  ks_enc_log.txt, kernel_stats_raw.csv,
  kernel_trace.csv                        rocprofv3 --kernel-trace --stats --output-format csv on bench.py
  pmc/{fetch,write,sq,l2}.csv             rocprofv3 --pmc passes on bench.py --chunks N --steps 1
  bench_mf.txt, ks_mf_log.txt,
  mf_kernel_stats_raw.csv                 the same two runs with --workload mf
(The decode summaries come from the same commands with --workload decode.)
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def kernel_class(name):
    if "encode_kernel" in name:
        return "encode_kernel"
    if "mf_scan" in name:
        return "mf_scan_kernel"
    if "rs_" in name:
        return "mf_radix_sort_kernels"
    if "rocprim" in name:
        return "rocprim_sorts_and_scans"
    if "mf_" in name:
        return "mf_table_kernels"
    return "other"


def short_name(name):
    """Kernel names without argument lists; library kernels by their rocprim entry point and key type."""
    m = re.search(r"(encode_kernel|decode_kernel<\w+>|(mf|rs)_\w+_kernel(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"wrapped_(\w+?)_config<[^;,]*[;,] *unsigned (long|int)", name)
    if m:
        return f"rocprim::{m.group(1)} <{'u64' if m.group(2) == 'long' else 'u32'} keys>"
    m = re.search(r"rocprim::\w+::detail::(\w+)", name)
    if m:
        return "rocprim::" + m.group(1)
    return name.split("(")[0][:80]


def bench_line(path):
    lines = [l for l in path.read_text(errors="replace").splitlines() if l.startswith('{"metric"')]
    return lines[-1]


def decode_summaries(run, prof, tag):
    """Bench line, decode_kernel row of the kernel stats, and the counters per output byte."""
    import ast
    (prof / f"{tag}_bench_decode.json").write_text(bench_line(run / "bench_decode.txt") + "\n")
    same = json.loads(bench_line(run / "ks_dec_log.txt"))
    rows = [r for r in csv.DictReader(open(run / "dec_kernel_stats_raw.csv")) if "decode_kernel" in r["Name"]]
    with open(prof / f"{tag}_decode_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload decode --no-cpu-baseline --no-verify  "
                f"({same['config']['streams_total']} streams, 1 warmup + 2 timed steps; MI355X; durations in ns; the run's "
                "other kernels belong to the encoder that prepares the streams and are left out)  -- the bench line of "
                f"this same run: value {same['value']} MB/s, roofline.kernel_ms {same['roofline']['kernel_ms']}\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            f.write(",".join([short_name(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs",
                                                                       "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
    out_bytes = same["config"]["raw_bytes_this_rank"]
    v = {}
    for d in ("fetch", "write", "sq", "l2"):
        for line in (run / "pmc_dec" / f"{d}.txt").read_text().splitlines():
            v.update(ast.literal_eval(line[line.index("{"):]))
    per = {k: round(x / out_bytes, 6 if k.endswith("SIZE") else 2) for k, x in v.items()}
    per = {(k + "_KB" if k.endswith("SIZE") else k): x for k, x in per.items()}
    per["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3)
    per["hbm_bytes_per_output_byte_reported"] = round((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / out_bytes, 2)
    res = {
        "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --workload decode --steps 1 "
                   "--warmup 0 --no-cpu-baseline --no-verify --no-e2e  (one run per counter group: FETCH_SIZE | WRITE_SIZE | "
                   "SQ_* | TCC_HIT_sum TCC_MISS_sum)",
        "workload": f"batched decode, {same['config']['streams_total']} streams of 4-64 KiB, {out_bytes:,} output bytes, MI355X",
        "notes": ["values are per output byte, summed over all waves of decode_kernel<false>",
                  "FETCH_SIZE / WRITE_SIZE in KB per output byte as rocprofv3 reports them (narrow accesses: "
                  "uncalibrated on gfx950)",
                  "SQ_* cycle counters in units of four cycles"],
        "decode_kernel_per_output_byte": per}
    (prof / f"{tag}_pmc_decode_{same['config']['streams_total'] // 1000}k_streams.json").write_text(
        json.dumps(res, indent=1) + "\n")
    print("decode", per)


def encode_stats(run, prof, tag):
    (prof / f"{tag}_bench_encode.json").write_text(bench_line(run / "bench_default.txt") + "\n")
    same = json.loads(bench_line(run / "ks_enc_log.txt"))
    rows = list(csv.DictReader(open(run / "kernel_stats_raw.csv")))
    keep = [r for r in rows if kernel_class(r["Name"]) != "other"]
    with open(prof / f"{tag}_encode_default_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-verify  (defaults: 4096 x "
                "1 MiB chunks, 1 warmup + 2 timed steps; MI355X; durations in ns; library kernel names shortened to "
                "their rocprim entry point)  -- the bench line of this same run: "
                f"value {same['value']} MB/s, ms_per_step {same['ms_per_step']:.1f}, roofline.kernel_ms "
                f"{same['roofline']['kernel_ms']} (= the encode_kernel average below)\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in keep:
            f.write(",".join([short_name(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs",
                                                                       "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")


def mf_stats(run, prof, tag):
    (prof / f"{tag}_bench_mf.json").write_text(bench_line(run / "bench_mf.txt") + "\n")
    same = json.loads(bench_line(run / "ks_mf_log.txt"))
    rows = [r for r in csv.DictReader(open(run / "mf_kernel_stats_raw.csv")) if kernel_class(r["Name"]) != "other"]
    scan = [r for r in rows if "mf_scan" in r["Name"]][0]
    with open(prof / f"{tag}_mf_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload mf --no-cpu-baseline --no-verify  "
                "(256 MiB, dict 1 MiB, 1 warmup + 2 timed steps; MI355X; durations in ns; library kernel names "
                "shortened to their rocprim entry point)  -- the bench line of this same run: "
                f"value {same['value']} MB/s, roofline.kernel_ms {same['roofline']['kernel_ms']} "
                f"(mf_scan_kernel average below: {float(scan['AverageNs']) / 1e6:.3f} ms)\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            f.write(",".join([short_name(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs",
                                                                       "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")


def main():
    run, tag, chunks = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    prof = Path(sys.argv[4]) if len(sys.argv) > 4 else ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    if (run / "bench_default.txt").exists():
        encode_stats(run, prof, tag)
    if (run / "bench_mf.txt").exists():
        mf_stats(run, prof, tag)
    if (run / "bench_decode.txt").exists():
        decode_summaries(run, prof, tag)
    if (run / "kernel_trace.csv").exists():
        timeline(run, prof, tag)
    if (run / "pmc" / "fetch.csv").exists():
        counters(run, prof, tag, chunks)
    if (run / "pmc_mf" / "fetch.csv").exists():
        counters(run, prof, tag, 256, sub="pmc_mf", name=f"{tag}_pmc_mf_256MiB.json",
                 cmd="bench.py --workload mf --steps 1 --warmup 0 --no-cpu-baseline --no-verify",
                 workload="match finder alone, one stream of 256 MiB, dict 1 MiB, fb 64, MI355X")


def timeline(run, prof, tag):
    trace = sorted(csv.DictReader(open(run / "kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
    ours = [r for r in trace if kernel_class(r["Kernel_Name"]) != "other"]
    t0 = int(ours[0]["Start_Timestamp"])
    with open(prof / f"{tag}_encode_default_timeline.txt", "w") as f:
        f.write("# kernel timeline of the same run (ms since the first finder kernel): E = encode_kernel launch (start "
                "end duration grid), scan = mf_scan_kernel, k = other finder kernels over 30 ms\n")
        for r in ours:
            b, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
            nm = r["Kernel_Name"]
            if "encode_kernel" in nm:
                f.write(f"E   {b:9.1f} {e:9.1f} {e - b:8.1f} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}\n")
            elif "mf_scan" in nm:
                f.write(f"  scan {b:9.1f} {e:9.1f} {e - b:8.1f}\n")
            elif e - b > 30:
                f.write(f"    k  {b:9.1f} {e:9.1f} {e - b:8.1f} {short_name(nm)}\n")


def counters(run, prof, tag, chunks, sub="pmc", name=None, cmd=None, workload=None):
    out = {}
    for d in ("fetch", "write", "sq", "l2"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls, seen = collections.Counter(), set()
        for row in csv.DictReader(open(run / sub / f"{d}.csv")):
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
    for k in ("encode_kernel", "mf_scan_kernel", "rocprim_sorts_and_scans", "mf_radix_sort_kernels", "mf_table_kernels"):
        if k not in out:
            continue  # (the finder's own sort only runs beside encode launches: not in a single-batch run)
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
        "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 "
                   + (cmd or f"bench.py --chunks {chunks} --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-extra")
                   + "  (one run per counter group: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_HIT_sum TCC_MISS_sum)",
        "workload": workload or f"encode, {chunks} chunks of 1 MiB (one finder batch, one encode launch), MI355X",
        "input_bytes": n_in,
        "notes": ["FETCH_SIZE/WRITE_SIZE are in KB as rocprofv3 reports them, summed over the kernel's dispatches",
                  "FETCH_SIZE on gfx950 reports half of the bytes of 16-byte-per-lane streaming reads; the reads here "
                  "are 12-byte entries and narrower, which are uncalibrated: treat the figures as a lower bound up to "
                  "a factor of two",
                  "SQ_* cycle counters are in units of four cycles, summed over all waves",
                  "LDS traffic of encode_kernel: SQ_INSTS_LDS wave instructions, most of them with one or a few "
                  "active lanes"],
        "kernels": rows}
    (prof / (name or f"{tag}_pmc_encode_{chunks}x1MiB.json")).write_text(json.dumps(res, indent=1) + "\n")
    for k, v in rows.items():
        print(k, v["launches"], "B/byte", v["hbm_bytes_per_input_byte_reported"], "L2", v["l2_hit_rate"],
              v["per_position"], "fetchGB", round(v["FETCH_SIZE_KB"] * 1024 / 1e9, 1), "writeGB",
              round(v["WRITE_SIZE_KB"] * 1024 / 1e9, 1))


if __name__ == "__main__":
    main()
