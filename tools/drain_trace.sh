#!/bin/bash
# tools/drain_trace.sh: kernel + copy timeline of ONE capture (throughput setting) and of the last passes of a K = 20 loop -- what is
# left behind the last hot kernel (the drain every K-step measurement pays once).  Output: gpurun_out/drain_trace.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/drain; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/drain_probe.py <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from dataclasses import replace
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = DevicePipeline(0, pipelined=True)
pipe.reserve(n, p)
st = pipe.stream(n, replace(p, write_bit_sample_pos=False), want_qad=True, want_pos=False)
for _ in range(6):
    for _ in range(30): st.push(iq)
    st.flush()
torch.cuda.synchronize()
for rep in range(3):
    time.sleep(0.002)
    t0 = time.perf_counter()
    for _ in range(20): st.push(iq)
    t1 = time.perf_counter()
    st.flush()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"K=20: pushes {1e6*(t1-t0):.0f} us, flush {1e6*(t2-t1):.0f} us, sync {1e6*(t3-t2):.0f} us, total {1e6*(t3-t0):.0f} us = {1e3*(t3-t0)/20:.4f} ms/step")
st.close()
PY
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o b -- python /tmp/drain_probe.py > $OUT/log.txt 2>&1)
grep "K=20" $OUT/log.txt > $R/gpurun_out/drain_trace.txt
python - >> $R/gpurun_out/drain_trace.txt <<PY
import csv, glob, re
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24]))
for f in glob.glob("$OUT/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + str(r.get("Size", ""))))
rows.sort()
hot = [i for i, r in enumerate(rows) if r[2].startswith("k_demod_runs")]
# the last K = 20 loop: the last 20 hot kernels; print from the 18th hot kernel on
i0 = hot[-3]
base = rows[i0][0]
print("last three passes of the last K = 20 loop (us from the third-last hot kernel's start):")
for r in rows[i0:]:
    print(f"  {(r[0]-base)/1000:9.1f} +{(r[1]-r[0])/1000:7.1f}  {r[2]}")
# the first pass of that loop
j0 = hot[-20]
print("first two passes of the loop:")
base = rows[j0][0]
for r in rows[j0 - 3:j0 + 14]:
    print(f"  {(r[0]-base)/1000:9.1f} +{(r[1]-r[0])/1000:7.1f}  {r[2]}")
PY
rm -rf $OUT
