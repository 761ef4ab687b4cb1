"""K = 80 pipelined steps of the headline capture as complex64, bits-only and int8 (ms per step): the A/B harness of round 6's tail experiments
(URHGPU_LIB=<tagged build> / URH_TUNE_<KEY>=<value> select the variant; profiles/r06_fir_pmc_account.txt)."""
import sys, time, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dataclasses import replace
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True, tuning={k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")})
pipe.reserve(n, p)
x8 = (iq * 64.0).round().clamp(-127, 127).to(torch.int8).contiguous()
out = {}
for name, x, dt, qad in (("f32", iq, np.float32, True), ("bits_only", iq, np.float32, False), ("int8", x8, np.int8, True)):
    st = pipe.stream(n, p, want_qad=qad, want_pos=False, dtype=dt)
    def run(k):
        for _ in range(k): st.push(x)
        st.flush()
    for _ in range(5): run(30)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(80); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 80 * 1e3)
    out[name] = round(min(ts), 4)
    st.close()
if "--sps10" in sys.argv:                                   # the 10-samples-per-symbol capture of bench.py's variants.sps10 (tolerance 1), K = 40
    del iq, x8
    iq10, _ = spec_fsk_capture(128, dev, first_segment=0, sps=10)
    p10 = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1, 10, 0.1, 8, False)
    st = pipe.stream(n, p10, want_qad=True, want_pos=False, dtype=np.float32)
    def run10(k):
        for _ in range(k): st.push(iq10)
        st.flush()
    for _ in range(5): run10(20)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run10(40); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    out["sps10"] = round(min(ts), 4)
    st.close()
print(json.dumps(out))
