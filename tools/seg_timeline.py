#!/usr/bin/env python3
"""tools/seg_timeline.py <rocprofv3 dir> [which]: kernels of the LAST single capture of tools/seg_probe.py (the last hot kernel that
is followed by an idle gap before the burst) relative to the start of its hot kernel; and 3 passes from the middle of the burst."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "urh::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("urh::", "").replace("void ", "")
    return n[:44]
hot = [i for i, r in enumerate(rows) if "k_demod_runs_bp" in r["Kernel_Name"]]
# gaps between consecutive hot kernels: singles are separated by > 1 ms
starts = [int(rows[i]["Start_Timestamp"]) for i in hot]
single_idx = [hot[j] for j in range(1, len(hot) - 1) if starts[j] - starts[j - 1] > 1_000_000 and starts[j + 1] - starts[j] > 1_000_000]
def show(i0, i1, title):
    t0 = int(rows[i0]["Start_Timestamp"])
    print(title)
    for r in rows[i0:i1]:
        print("  %9.1f %9.1f  %7.1f us  q=%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
if single_idx:
    i = single_idx[-1]
    nxt = [h for h in hot if h > i]
    show(i, nxt[0] if nxt else len(rows), "== one capture (idle machine)")
mid = hot[len(hot) - 8] if len(hot) > 10 else hot[-1]
end = hot[len(hot) - 6] if len(hot) > 10 else len(rows)
# every kernel that STARTS between the starts of two hot kernels in the burst, two passes' worth
show(mid, end, "== burst: two passes (everything that started between three consecutive hot kernels)")
