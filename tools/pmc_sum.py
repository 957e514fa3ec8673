"""Sums rocprofv3 counter_collection.csv rows per kernel: python tools/pmc_sum.py DIR [substring] [divisor]"""
import collections
import csv
import glob
import sys

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    print(k, {a: round(b / div, 2) for a, b in v.items()})
