#!/usr/bin/env python3
"""Costas loop on the config-5 capture (1 GiB 4-PSK) with a library built for another chunk length: time per call, stats,
and the demodulated signal's checksum (must be the same for every build).  usage: costas_chunk_ab.py <liburhgpu variant .so>"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["URHGPU_LIB"] = sys.argv[1]
import torch
from urh_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1] if hasattr(_lib, "LIB_PATH") else None
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_psk_capture
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = spec_psk_capture(128, dev)
p = DemodParams("PSK", 2, 0.2, 0.0, 1.5, 5, 100, 0.1, 8, True)
for _ in range(3):
    q = pipe.afp_demod(iq, p)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    q = pipe.afp_demod(iq, p)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
h = hashlib.sha1(q[1:].cpu().numpy().tobytes()).hexdigest()[:16]
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "default", "costas ms", round(sorted(ts)[2], 3), "stats", pipe.ctx.costas_stats(), "sha1", h)
