#!/usr/bin/env python3
"""Timeline of the urh:: kernels of a rocprofv3 --kernel-trace run: tools/show_overlap.py <dir> [n_rows]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "urh::" in r["Kernel_Name"] or "nccl" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = max(0, len(rows) // 2 - nshow // 2)
for r in rows[skip:skip + nshow]:
    print("%10.1f %10.1f  q=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"),
                                      r["Kernel_Name"][:60]))
