"""Sums of rocprofv3 --pmc counters per kernel: python scripts/pmc_kernel.py <counter_collection.csv> [kernel substring]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k in agg:
    print(k[:110], "dispatches", len(n[k]))
    for c, v in sorted(agg[k].items()):
        print("   %-32s %16.0f  per dispatch %14.0f" % (c, v, v / len(n[k])))
