#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3e3; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o e -- python $R/tools/est_probe.py --no-psk > $OUT/log.txt 2>&1)
python - <<PY
import csv,re,glob
rows=[]
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
for f in glob.glob("$OUT/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Size","")))
rows.sort()
def short(n):
    if n.startswith("COPY"): return n
    m=re.search(r"(k_[a-z_0-9]+)",n); return m.group(1) if m else n[:28]
idx=[i for i,r in enumerate(rows) if "k_demod_runs_bp" in r[2]]
start=idx[-1]
t0=rows[start][0]
prev_end=None
for s,e,n in rows[start:start+80]:
    gap = (s-prev_end)/1000 if prev_end else 0
    print(f"{(s-t0)/1000:8.1f} +{(e-s)/1000:7.1f}  gap {gap:6.1f}  {short(n)}")
    prev_end=max(prev_end or 0,e)
PY
