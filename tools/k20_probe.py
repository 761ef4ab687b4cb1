"""tools/k20_probe.py: the driver's measurement in small -- K = 20 pipelined steps of the headline capture through urhgpu_stream_* (push x K, flush,
synchronize), median and minimum of 15 repetitions behind a clock ramp.  URHGPU_LIB selects an A/B build, URH_TUNE_<KEY>=<value> a tuning key."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True, tuning={k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")})
pipe.reserve(n, p)
st = pipe.stream(n, p, want_qad=True, want_pos=False)
def run(k):
    for _ in range(k): st.push(iq)
    st.flush()
for _ in range(8): run(20)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    run(100)                                       # clocks
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
ts.sort()
print(json.dumps({"k20_median": round(ts[len(ts) // 2], 4), "k20_min": round(ts[0], 4)}))
