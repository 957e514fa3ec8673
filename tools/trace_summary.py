"""Timeline of an encode benchmark step from a rocprofv3 kernel trace: per finder batch the
time spent in sorts (library kernels), table kernels and the scan, and the encode launches.

    python tools/trace_summary.py gpurun_out/<run>/kernel_trace.csv
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
enc = [r for r in rows if "encode_kernel" in r["Kernel_Name"]]
# the last step: kernels after the second-to-last group of encode launches
nl = len(enc) // 2 if len(enc) >= 2 else len(enc)
start_idx = rows.index(enc[-nl]) if enc else 0
# walk back to the first finder kernel of that step
i = start_idx
while i > 0 and "encode_kernel" not in rows[i - 1]["Kernel_Name"]:
    i -= 1
step = rows[i:]
t0 = int(step[0]["Start_Timestamp"])
acc = {"sort": 0.0, "table": 0.0}
batch_start = None
for r in step:
    name = r["Kernel_Name"]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if "encode_kernel" in name:
        print(f"E      {s:9.1f} {e:9.1f} {e - s:8.1f}  grid {r.get('Grid_Size', '?')}")
    elif "mf_scan" in name:
        print(f"  batch {batch_start:9.1f}: sorts {acc['sort']:7.1f}  tables {acc['table']:7.1f}  scan {s:9.1f} {e:9.1f} {e - s:8.1f}  (batch {e - batch_start:7.1f})")
        acc = {"sort": 0.0, "table": 0.0}
        batch_start = None
    elif "rocprim" in name or "mf_" in name or "rs_" in name or "copyBuffer" in name:
        if batch_start is None:
            batch_start = s
        acc["table" if "mf_" in name else "sort"] += e - s
