#!/usr/bin/env python3
"""OOK IQ->bits only, a few passes over a 1 GiB capture (for rocprofv3 --stats; developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture

dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, dev, seed=1)
pa = DemodParams("ASK", 1, 0.02, 0.3, 1.0, 5, 100)
for _ in range(12):
    r = pipe.iq_to_bits(iq, pa, want_qad=True)
torch.cuda.synchronize()
print(r.host_counts())
