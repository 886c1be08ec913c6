#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name: pmc_summary.py <dir> [name-substring]"""
import collections, csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/*/*_counter_collection.csv")[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if pat not in k:
        continue
    k = k[:70]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k in tot:
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, v in sorted(tot[k].items()):
        print(f"   {c:34s} {v / n:16.0f}")
