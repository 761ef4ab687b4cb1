#!/usr/bin/env python3
"""Round 4: latency of ONE resident capture (push + flush on an idle GPU) under tuning dicts given as k=v,k=v arguments"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, bool(int(os.environ.get("WANT_POS", "0"))))
for spec in sys.argv[1:]:
    tun = dict((k, int(v)) for k, v in (kv.split("=") for kv in spec.split(","))) if spec != "default" else {}
    pipe = DevicePipeline(0, pipelined=True, tuning=tun)
    pipe.reserve(n, p)
    st = pipe.stream(n, p, want_qad=True, want_pos=p.write_bit_sample_pos)
    for _ in range(3):
        st.push(iq); st.flush()
    for _ in range(150):
        st.push(iq)
    st.flush(); torch.cuda.synchronize()
    one = []
    for _ in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.push(iq); st.flush()
        one.append((time.perf_counter() - t0) * 1e3)
    one.sort()
    print(f"{spec:70s} min {one[0]:.4f} median {one[8]:.4f} ms", flush=True)
    st.close(); del st, pipe
