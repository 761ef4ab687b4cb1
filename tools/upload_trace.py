#!/usr/bin/env python3
"""tools/upload_trace.py <rocprofv3 dir>: memory copies and kernels of the LAST upload pass, relative to its first copy"""
import csv, glob, os, sys
d = sys.argv[1]
def load(pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
cp = [r for r in load("*memory_copy_trace.csv")]
kn = [r for r in load("*kernel_trace.csv") if "urh::" in r["Kernel_Name"]]
ev = []
for r in cp:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?") ))
for r in kn:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("urh::", "").replace("void ", "")[:40]))
ev.sort()
big = [e for e in ev if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2].upper() and e[1] - e[0] > 200_000]
if not big:
    print("no big H2D copies found; directions:", sorted({e[2] for e in ev if e[2].startswith("COPY")})); sys.exit()
# last pass: walk back from the last big copy while gaps are < 5 ms
i = len(big) - 1
while i > 0 and big[i][0] - big[i - 1][1] < 5_000_000: i -= 1
t0 = big[i][0]
for s, e, n in ev:
    if s >= t0 - 100_000:
        print("%10.1f %10.1f %9.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
