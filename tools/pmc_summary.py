"""Sums rocprofv3 --pmc counter_collection csv per (kernel, counter): python tools/pmc_summary.py <dir> [kernel substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if sub not in k:
            continue
        a = acc[(k[:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-60s %-34s %16.0f  per launch (%d launches)" % (k, c, v / n, n))
