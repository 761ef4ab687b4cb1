"""Probe: N independent contexts / streams, passes dealt round-robin (GPU box).  python tools/multi_stream_probe.py N [steps]"""
import sys
import time

sys.path.insert(0, ".")
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture

N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
iq, _ = fsk_capture(128, dev, seed=1234)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipes = [DevicePipeline(0) for _ in range(N)]
streams = [torch.cuda.Stream(dev) for _ in range(N)]
for k in range(N):
    pipes[k].reserve(iq.shape[0], p)
    with torch.cuda.stream(streams[k]):
        for _ in range(2):
            r = pipes[k].iq_to_bits(iq, p, want_qad=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    k = i % N
    with torch.cuda.stream(streams[k]):
        r = pipes[k].iq_to_bits(iq, p, want_qad=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"N={N} steps={steps} ms/step={dt / steps * 1e3:.4f} value={iq.shape[0] * steps / dt / 1e6:.0f} Msamples/s counts={r.host_counts()}")
