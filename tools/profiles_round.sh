#!/bin/bash
# The round's record for profiles/: tools/profiles_round.sh <tag>  (one gpurun call, about ten minutes)
#   bench line (live PMC traffic, variants), kernel stats + timeline of the pipelined and the un-pipelined loop, PMC of the hot kernel,
#   timeline of a single capture and of an uploaded one, FIR and Costas kernel stats + PMC, estimator kernels, sample types, the 1-rank
#   sharded line with the FIR-halo record.  Copy what is to be judged from gpurun_out/<tag>/ into profiles/.
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
G=gpurun_out/$TAG; mkdir -p $G
python bench.py --steps 20 --warmup 5 > $G/bench.json 2> $G/bench.err
tail -c 300 $G/bench.err
for mode in pipelined unpipelined; do
  OUT=$R/$G/stats_$mode; mkdir -p $OUT
  if [ $mode = pipelined ]; then A="--no-d2h --no-reference-loop"; else A="--no-d2h --no-pipeline"; fi
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra --no-upload --no-pmc $A > $OUT/log.txt 2>&1)
  python tools/timeline.py $OUT --passes 12 > $G/timeline_$mode.txt 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "urh::" $f) > $G/fsk_1gib_${mode}_kernel_stats.csv
  find $OUT -name "*kernel_trace.csv" -delete
done
bash tools/prof_pmc.sh $TAG --no-upload --no-pmc > $G/pmc_files.txt 2>&1
python tools/prof_collect.py gpurun_out/prof_$TAG $G/fsk_1gib > $G/pmc_summary.txt 2>&1
find gpurun_out/prof_$TAG -name "*.csv" -size +2M -delete
# one capture on an idle GPU (latency setting: segments), one uploaded capture: kernel (+ copy) timelines
POLICY=1 bash tools/seg_prof.sh "7 0" > /dev/null 2>&1; cp gpurun_out/seg_7_0.txt $G/single_capture_timeline.txt
OUT=$R/$G/up; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o b -- python $R/tools/upload_probe.py 4 > $OUT/log.txt 2>&1)
grep "upload in\|bare pinned" $OUT/log.txt > $G/upload_timeline.txt
python tools/upload_trace.py $OUT/trace | head -60 >> $G/upload_timeline.txt 2>&1; rm -rf $OUT/trace
# FIR (64 taps) and Costas (order 4) at 1 GiB: kernel stats, then the SQ / GRBM counters
for what in fir costas; do
  OUT=$R/$G/$what; mkdir -p $OUT
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $R/tools/${what}_only.py > $OUT/stats.log 2>&1)
  f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep "urh::" $f) > $G/${what}_1gib_kernel_stats.csv
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "k_fir|k_costas" --output-format csv -d $OUT/pmc$i -o b -- python $R/tools/${what}_only.py > $OUT/pmc$i.log 2>&1)
  done
  python - > $G/${what}_1gib_pmc.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("urh::", "").replace("(anonymous namespace)::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} mean {sum(v) / len(v):16.1f}  launches {len(v)}")
PY
  find $OUT -name "*.csv" -size +1M -delete
done
bash tools/est_prof.sh ${TAG}_est > /dev/null 2>&1
for part in ook psk; do echo "== $part"; grep -v "^[WE]2026\|rocprofv3" gpurun_out/${TAG}_est/log_$part.txt | tail -1 | cut -c1-700; head -24 gpurun_out/${TAG}_est/kernels_$part.txt; done > $G/estimate_kernels.txt 2>&1
python tools/dtype_probe.py 2>&1 | grep -v amdgpu.ids > $G/dtypes.txt
python tools/deviation_probe.py 2>&1 | grep -v amdgpu.ids > $G/deviation.txt
# the sharded path on a 1-rank RCCL group, with its self-check and the FIR-halo record
URH_BENCH_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 > $G/bench_sharded_1rank.json 2> $G/bench_sharded_1rank.err
python - <<PY
import json
d = json.loads(open("$G/bench.json").read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
print("ms/step", d["ms_per_step"], "kernel", r["kernel_ms"], "frac", r["frac"], "e2e", r.get("end_to_end_frac"), "traffic", r.get("traffic"), r.get("traffic_source"))
print("single", c.get("single_capture_incl_compact_d2h_ms"), "with pos", c.get("ms_per_step_with_device_positions"), "h2d", c.get("bare_pinned_h2d_ms"), c.get("h2d_inclusive_ms"))
print("variants", json.dumps(c.get("variants"))[:900])
print(c.get("configs2_ook_fir"), c.get("configs4_psk_costas"))
PY
tail -3 $G/pmc_summary.txt; tail -3 $G/timeline_pipelined.txt; cat $G/dtypes.txt
