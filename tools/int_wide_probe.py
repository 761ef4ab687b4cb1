"""tools/int_wide_probe.py: K = 80 pipelined steps of an int8 capture of the headline signal at +-20 kHz (narrow) and at +-100 kHz (wide) through
urhgpu_stream_* -- the stream probes its captures and picks the hot kernel's instantiation (k_wide_probe, RunArgs::wide_int).  Under rocprofv3
--kernel-trace the kernel names show which one ran (last template argument)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
dev = torch.device("cuda", 0)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True)
out = {}
for name, dev_hz in (("narrow_20kHz", 20e3), ("wide_100kHz", 100e3), ("narrow_again", 20e3)):
    iq, _ = fsk_capture(128, dev, seed=1234, deviation_hz=dev_hz)
    x8 = (iq * 64.0).round().clamp(-127, 127).to(torch.int8).contiguous()
    n = x8.shape[0]
    del iq
    if "st" not in globals():
        pipe.reserve(n, p)
        st = pipe.stream(n, p, want_qad=True, want_pos=False, dtype=np.int8)
    def run(k):
        for _ in range(k): st.push(x8)
        st.flush()
    for _ in range(4): run(30)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(80); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 80 * 1e3)
    out[name] = round(min(ts), 4)
st.close()
print(json.dumps(out))
