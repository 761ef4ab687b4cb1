#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
cat > /tmp/up_one.py <<'P'
import os, sys, time
sys.path.insert(0, os.environ["R"])
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
iq, _ = spec_fsk_capture(128, torch.device("cuda", 0)); n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pinned = torch.empty(iq.shape, dtype=iq.dtype, pin_memory=True); pinned.copy_(iq); dst = torch.empty_like(iq)
mode = os.environ.get("MODE", "flush")
for spec in sys.argv[1:]:
    tun = dict((k, int(v)) for k, v in (kv.split("=") for kv in spec.split(",")))
    pipe = DevicePipeline(0, pipelined=True, tuning=tun); pipe.reserve(n, p)
    st = pipe.stream(n, p, want_qad=True, want_pos=False)
    st.push(iq); st.flush(); t = []
    for _ in range(4):
        dst.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st.push_upload(pinned, dst)
        if mode == "devsync":
            torch.cuda.synchronize()
        elif mode == "busy":
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 0.0195: pass
        st.flush(); t.append((time.perf_counter() - t0) * 1e3)
    print(mode, spec, "min %.3f all %s" % (min(t), [round(x, 2) for x in t]), flush=True)
    st.close(); del st, pipe
P
export R
for m in flush devsync busy; do MODE=$m timeout 100 python /tmp/up_one.py upload_pieces=16,stream_spin=0 upload_pieces=8,stream_spin=0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4s2_up_modes.txt; done
cat gpurun_out/r4s2_up_modes.txt
OUT=$R/gpurun_out/r4up16b; mkdir -p $OUT
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o b -- python /tmp/up_one.py upload_pieces=16,stream_spin=0 > $OUT/log.txt 2>&1)
tail -1 $OUT/log.txt
python tools/r4_upload_trace.py $OUT/trace > gpurun_out/r4s2_up16b_trace.txt 2>&1; rm -rf $OUT/trace
grep -n "COPY\|k_seg_gate" gpurun_out/r4s2_up16b_trace.txt | head -40
