#!/bin/bash
# dynamic instruction counts of the hot kernel against the FSK deviation: rocprofv3 --pmc over tools/deviation_probe.py <kHz>
# usage: bash tools/deviation_pmc.sh <out.txt> [kHz ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
DST=${1:-/dev/stdout}; shift
for khz in ${@:-20 50 100}; do
  OUT=$(mktemp -d); i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_demod_runs_bp" --output-format csv -d $OUT/p$i -o d -- python $R/tools/deviation_probe.py $khz > $OUT/log$i.txt 2>&1)
  done
  python3 - $OUT $khz >> $DST <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = (1 << 27) // 128
m = {c: sum(v) / len(v) for c, v in acc.items()}
print(f"deviation {sys.argv[2]} kHz: per row of 128 samples VALU {m.get('SQ_INSTS_VALU', 0) / rows:.1f}  SALU {m.get('SQ_INSTS_SALU', 0) / rows:.1f}  LDS {m.get('SQ_INSTS_LDS', 0) / rows:.2f};  "
      f"VALU issue {m.get('SQ_ACTIVE_INST_VALU', 0) * 4 / max(1024 * m.get('GRBM_GUI_ACTIVE', 1) / 8, 1):.3f};  GRBM_GUI_ACTIVE {m.get('GRBM_GUI_ACTIVE', 0):.0f}")
PY
  rm -rf $OUT
done
