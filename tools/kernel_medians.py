"""Median / 10th / 90th percentile duration per kernel from a rocprofv3 kernel trace (…_kernel_trace.csv): the averages of --stats hide the
first calls and the rare long ones.   usage: python tools/kernel_medians.py <kernel_trace.csv> [name filter]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in rows:
    if flt in r["Kernel_Name"]:
        d[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-66s %6s %8s %8s %8s %9s" % ("kernel", "calls", "median", "p10", "p90", "max"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-66s %6d %8.1f %8.1f %8.1f %9.1f" % (k, len(v), statistics.median(v), v[len(v) // 10], v[len(v) * 9 // 10], v[-1]))
