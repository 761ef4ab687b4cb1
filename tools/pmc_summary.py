#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs written by tools/prof_pmc.sh: per-kernel average duration and counters."""
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    print("==", f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.1f} pct={r['Percentage']}")
for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k:60s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
