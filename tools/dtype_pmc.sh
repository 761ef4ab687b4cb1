#!/bin/bash
# dynamic instruction counts of the hot kernel per sample type: rocprofv3 --pmc over tools/dtype_probe.py
# usage: bash tools/dtype_pmc.sh <out.txt>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(mktemp -d)
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_demod_runs_bp" --output-format csv -d $OUT/p$i -o d -- python $R/tools/dtype_probe.py > $OUT/log$i.txt 2>&1)
done
python3 - $OUT > ${1:-/dev/stdout} <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = (1 << 27) // 128
names = {"<0, 4, 1, true, true, 1, false>": "complex64", "<0, 2, 1, true, true, 1, false>": "int16", "<0, 0, 1, true, true, 1, false>": "int8"}
print("# k_demod_runs_bp, 2^27 samples = %d rows of 128: wave-instructions per launch (mean over the launches of tools/dtype_probe.py) and per row" % rows)
for k, cs in sorted(acc.items()):
    label = next((v for s, v in names.items() if s in k), None)
    if not label:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"{label:10s} " + "  ".join(f"{c} {m[c]:.4g}" for c in sorted(m)))
    if "SQ_INSTS_VALU" in m:
        print(f"{'':10s} per row: VALU {m['SQ_INSTS_VALU'] / rows:.1f}  SALU {m.get('SQ_INSTS_SALU', 0) / rows:.1f}  VMEM rd {m.get('SQ_INSTS_VMEM_RD', 0) / rows:.2f} wr {m.get('SQ_INSTS_VMEM_WR', 0) / rows:.2f}  LDS {m.get('SQ_INSTS_LDS', 0) / rows:.2f}")
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        print(f"{'':10s} VALU issue: SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = {m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * m['GRBM_GUI_ACTIVE'] / 8):.3f}  (un-pipelined passes: all 256 CUs)")
PY
rm -rf $OUT
