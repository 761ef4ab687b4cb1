#!/bin/bash
# tools/icache_ab.sh: SQC_ICACHE_REQ / MISSES / HITS of the headline instantiation for the default build and liburhgpu_nowide.so
# (tools/quick_tag.sh nowide demod_runs.hip -DURH_NO_WIDE=1), over tools/ab_variants.py
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_CYCLES_[A-Z]*" | sort -u | tr '\n' ' '; echo
for lib in liburhgpu.so liburhgpu_nowide.so; do
  OUT=$(mktemp -d)
  URHGPU_LIB=$GRAFT_REPO_ROOT/urh_amd/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_WAVES --kernel-include-regex "k_demod_runs_bp" --output-format csv -d $OUT -o d -- python $GRAFT_REPO_ROOT/tools/ab_variants.py > $OUT/log.txt 2>&1
  python3 - $OUT $lib <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ILi0ELi4ELi1ELb1ELb1" in r["Kernel_Name"] or "<0, 4, 1, true, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", {k: len(v) for k, v in acc.items()})
PY
  tail -1 $OUT/log.txt | cut -c1-200
done
