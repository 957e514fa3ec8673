"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files.

    python tools/pmc_kernels.py gpurun_out/<run>/{fetch,write,sq,l2}.csv
"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(encode_kernel|decode_kernel|mf_\w+_kernel)", name)
    if m:
        return m.group(1)
    if "rocprim" in name:
        return "rocprim"
    return name.split("(")[0][:40]


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (path, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
    for row in csv.DictReader(open(path)):
        pass
for k, d in sorted(tot.items()):
    print(k, {c: f"{v:.4g}" for c, v in sorted(d.items())})
