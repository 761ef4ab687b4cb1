#!/bin/bash
# PMC counters of the estimator kernels at full size (center statistics of the config-3 OOK capture and of the config-5 PSK signal):
# VALU instructions, waves, busy cycles, HBM bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, as the guide prescribes) per launch.
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_estpmc; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_me_" --output-format csv -d $OUT/pmc$i -o e -- python $R/tools/est_probe.py > $OUT/pmc$i.log 2>&1
done
python - > $R/gpurun_out/${TAG}_estimator_pmc.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("urh::", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("per launch means (all launches of the probe: OOK config-3 centers + plateaus, PSK config-5 detect_center)")
for name in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    n = {k: len(v) for k, v in acc[name].items()}
    hbm = c.get("FETCH_SIZE", 0) * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024
    print(f"{name:22s} launches {max(n.values()):4d}  VALU wave-instr {c.get('SQ_INSTS_VALU', 0):12.0f}  waves {c.get('SQ_WAVES', 0):9.0f}  LDS instr {c.get('SQ_INSTS_LDS', 0):10.0f}  "
          f"VMEM_RD {c.get('SQ_INSTS_VMEM_RD', 0):10.0f}  busy cycles (GRBM, all XCDs) {c.get('GRBM_GUI_ACTIVE', 0):10.0f}  HBM bytes {hbm / 1e6:9.1f} MB")
PY
cat $R/gpurun_out/${TAG}_estimator_pmc.txt
find $OUT -name "*.csv" -size +1M -delete
