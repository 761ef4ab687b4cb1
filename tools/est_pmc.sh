#!/bin/bash
# PMC counters of the estimator kernels (GPU box): tools/est_pmc.sh TAG REGEX [ook|psk]   -> gpurun_out/TAG_pmc/summary.txt
# (counters in their own passes with --kernel-trace only, MI355X_MICROARCH.md)
TAG=$1; RX=${2:-k_me_hist}; PART=${3:-psk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_pmc; mkdir -p $OUT
other=$([ $PART = ook ] && echo --no-psk || echo --no-ook)
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc$i -o b -- python $R/tools/est_probe.py $other > $OUT/pmc$i.log 2>&1
done
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k:60s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
cat $OUT/summary.txt
