#!/bin/bash
# HBM traffic of the hot kernel when it shares the machine with the previous pass's tail (pipelined, device-only) against passes one after the other
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3t; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for mode in piped alone; do
  if [ $mode = piped ]; then A="--no-d2h --no-reference-loop"; else A="--no-d2h --no-pipeline"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_demod" --output-format csv -d $OUT/${mode}_$c -o b -- python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extra $A > $OUT/${mode}_$c.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
for mode in ("piped", "alone"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (mode, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_demod_runs_bp" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        vals = vals[len(vals) // 2:]          # the timed steps (second half of the launches)
        tot[c] = sum(vals) / max(len(vals), 1)
        print(mode, c, "mean KiB", round(tot[c], 1), "launches", len(vals))
    print(mode, "HBM bytes per launch (FETCH x 2 + WRITE):", round((tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024 / 1e9, 4), "GB  (algorithmic 1.6106)")
PY
find $OUT -name "*.csv" -size +1M -delete
