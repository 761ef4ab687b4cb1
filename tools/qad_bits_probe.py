#!/usr/bin/env python3
"""qad -> pulse table -> bits on a 2^27-sample demodulated signal (config 3 / 5's last stage): wall time per call; kernel times under rocprofv3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_ook_capture
pipe = DevicePipeline(0)
iq, _ = spec_ook_capture(128, torch.device("cuda", 0))
p = DemodParams("ASK", 1, 0.02, 0.32, 1.0, 5, 100, 0.1, 8, True)
qad = pipe.afp_demod(iq, p)
del iq
for _ in range(100): r = pipe.qad_to_bits(qad, p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): r = pipe.qad_to_bits(qad, p); torch.cuda.synchronize()
print("qad_to_bits ms per call", round((time.perf_counter() - t0) / 20 * 1e3, 4), r.host_counts())
