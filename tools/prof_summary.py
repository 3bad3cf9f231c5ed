"""Condenses rocprofv3 csv output (kernel stats + PMC passes) of tools/prof.sh into one text summary."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    """`void ns::(anonymous namespace)::kernel<7, 8>(args...)` -> `kernel<7, 8>`."""
    name = re.sub(r"^void\s+", "", name)
    name = name.split("(sadbm_detail")[0].split("(unsigned")[0]
    name = name.replace("(anonymous namespace)::", "").replace("sadbm_detail::", "").replace("strip::", "")
    return name.split("(")[0][:60]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-40s calls=%s total_ns=%s avg_ns=%s pct=%s" % (short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"),
                                                              row.get("AverageNs"), row.get("Percentage")))
print("== PMC (per-dispatch mean over dispatches of each kernel) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s mean=%.6g  n=%d" % (c, sum(v) / len(v), len(v)))
