"""Pipelined steps per sample type against the number of CUs per XCD the hot kernel leaves to the tail (tuning key
hot_cus_removed_per_xcd; shipped: 4): python tools/dtype_mask_sweep.py [removed ...]
The int16 / int8 / bits-only steps are bound by the tail chain of the pass before: does a wider tail area pay for them?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
removed = [int(a) for a in sys.argv[1:]]
if not removed:                                             # one process per value (contexts do not share a GPU well), the shipped value at both ends
    import subprocess
    for r in (4, 2, 3, 5, 6, 8, 4):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(r)], check=False)
    sys.exit(0)
K = int(os.environ.get("URH_SWEEP_K", "40"))
print(f"# {torch.cuda.get_device_name(0)}; K = {K}; ms per pipelined step")
cases = (("float32", torch.float32, np.float32, 1.0, True), ("bits_only", torch.float32, np.float32, 1.0, False),
         ("int16", torch.int16, np.int16, 8192.0, True), ("int8", torch.int8, np.int8, 64.0, True))
xs = {name: (iq if ndt is np.float32 else (iq * scale).round().to(tdt).contiguous()) for name, tdt, ndt, scale, _ in cases}
print("removed " + " ".join(f"{c[0]:>10s}" for c in cases))
for r in removed:
    tuning = {k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")}       # further keys: URH_TUNE_<KEY>=<value>
    tuning["hot_cus_removed_per_xcd"] = r
    pipe = DevicePipeline(0, pipelined=True, tuning=tuning)
    pipe.reserve(n, p)
    row = []
    for name, tdt, ndt, scale, wq in cases:
        x = xs[name]
        st = pipe.stream(n, p, want_qad=wq, want_pos=False, dtype=ndt)
        def run(k):
            for _ in range(k):
                st.push(x)
            st.flush()
        for _ in range(6):
            run(30)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / K * 1e3)
        st.close()
    print(f"{r:7d} " + " ".join(f"{v:10.4f}" for v in row), flush=True)
    del pipe
