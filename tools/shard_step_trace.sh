#!/bin/bash
# tools/shard_step_trace.sh [fir|plain]: kernel timeline of the sharded engine's step (tools/shard_step_probe.py) -> gpurun_out/shard_step_<what>.txt
W=${1:-fir}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/shard_tl_$W; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o e -- python $R/tools/shard_step_probe.py $W 12 > $OUT/log.txt 2>&1)
python3 - $OUT > $R/gpurun_out/shard_step_$W.txt <<PY
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("urh::", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"[(<].*", "", n.replace("void ", ""))[:40], r.get("Queue_Id", "")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[-12:] + " " + str(r.get("Size", "")), "-"))
rows.sort()
what = "$W"
marker = "k_fir_fast" if what.startswith("fir") else "k_demod_runs_bp"
marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
# the first timed loop of the probe: steps 10 .. 10 + 12 of the marker kernel (fir) / the last 14 (plain); print three steps from its middle
i0, i1 = (marks[14], marks[17]) if what == "fir" else (marks[-8], marks[-5])
base = rows[i0][0]
print(f"three steps: {(rows[i1][0] - base) / 3000:.1f} us per step (marker kernel start to start)")
for r in rows[i0:i1]:
    print(f"  {(r[0] - base) / 1000:9.1f} +{(r[1] - r[0]) / 1000:7.1f}  q{r[3]:>3s}  {r[2]}")
PY
tail -4 $OUT/log.txt | cut -c1-400 >> $R/gpurun_out/shard_step_$W.txt
rm -rf $OUT
