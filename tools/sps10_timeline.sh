cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sps10_tl; rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o e -- python $GRAFT_REPO_ROOT/tools/ab_variants.py --sps10 > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/timeline.py $OUT --passes 6 > $GRAFT_REPO_ROOT/gpurun_out/sps10_timeline.txt 2>&1
tail -12 $GRAFT_REPO_ROOT/gpurun_out/sps10_timeline.txt
python3 - $OUT <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction",""), int(r.get("Size",0) or 0)))
rows.sort()
for r in rows[-12:]: print("copy", r[2][-14:], r[3], "bytes", (r[1]-r[0])/1e3, "us", round(r[3]/max(r[1]-r[0],1),2), "GB/s")
PY
find $OUT -name "*.csv" -size +1M -delete
