#!/bin/bash
# per-sample-type and per-deviation step times + the hot kernel's own time per instantiation (rocprofv3): profiles/r03*_dtypes.txt, *_deviation.txt
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python $R/tools/dtype_probe.py 2>/dev/null > $OUT/dtypes.txt
python $R/tools/deviation_probe.py 2>/dev/null > $OUT/deviation.txt
for t in dtype deviation; do
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$t -o k -- python $R/tools/${t}_probe.py > /dev/null 2>&1)
  python - >> $OUT/$([ $t = dtype ] && echo dtypes || echo deviation).txt <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace_$t/k_kernel_stats.csv")))
print("hot kernel per instantiation (rocprofv3 --kernel-trace --stats of this probe; <SRC, DT, MOD, ...>; DT 0..4 = int8, uint8, int16, uint16, float32):")
for r in rows:
    if "k_demod_runs" in r["Name"]:
        print(f'  {r["Name"].replace("urh::", "")[:70]:70s} calls {r["Calls"]:>5} avg_us {float(r["AverageNs"]) / 1e3:8.1f} min_us {float(r["MinNs"]) / 1e3:8.1f}')
PY
  find $OUT/trace_$t -name "*_trace.csv" -delete
done
cat $OUT/dtypes.txt $OUT/deviation.txt
