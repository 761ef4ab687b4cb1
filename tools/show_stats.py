#!/usr/bin/env python3
"""Print the urh:: kernels of a rocprofv3 --stats directory: tools/show_stats.py <dir>"""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if "urh::" in r["Name"]:
            print("%-105s %5s %10.1f us" % (r["Name"][:105], r["Calls"], float(r["AverageNs"]) / 1000))
            tot += float(r["TotalDurationNs"])
    print("total urh:: kernel time %.1f us" % (tot / 1000))
