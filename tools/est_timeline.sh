#!/bin/bash
# tools/est_timeline.sh [ook|psk]: kernels and copies of ONE estimate_dev (ook) / detect_center_dev (psk) call at 1 GiB in launch order, with
# start offsets and durations (rocprofv3 --kernel-trace --memory-copy-trace of tools/est_probe.py).  Output: gpurun_out/est_timeline_<part>.txt
PART=${1:-ook}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/est_tl_$PART; rm -rf $OUT; mkdir -p $OUT
other=$([ $PART = ook ] && echo --no-psk || echo --no-ook)
(cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o e -- python $R/tools/est_probe.py $other > $OUT/log.txt 2>&1)
python3 - $PART > $R/gpurun_out/est_timeline_$PART.txt <<PY
import csv, glob, re, sys
part = sys.argv[1]
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("urh::", "").replace("(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", n)[:60]))
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + str(r.get("Size", ""))))
rows.sort()
first = "void k_demod_runs_bp<0, 4, 0" if part == "ook" else "k_me_first"
marks = [i for i, r in enumerate(rows) if r[2].startswith(first)]
# the third-last call (the last one carries per-stage synchronisations)
i0, i1 = marks[-3], marks[-2]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 30000: i0 -= 1      # ops queued right before the marker kernel belong to the call
base = rows[i0][0]
print(f"one call: {len(rows[i0:i1])} operations, {(rows[i1 - 1][1] - base) / 1000:.1f} us from the first start to the last end")
prev_end = base
for r in rows[i0:i1]:
    print(f"  {(r[0] - base) / 1000:9.1f} +{(r[1] - r[0]) / 1000:7.1f}  gap {max(0, r[0] - prev_end) / 1000:6.1f}  {r[2]}")
    prev_end = max(prev_end, r[1])
PY
tail -2 $OUT/log.txt | cut -c1-600 >> $R/gpurun_out/est_timeline_$PART.txt
rm -rf $OUT
