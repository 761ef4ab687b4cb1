#!/usr/bin/env python3
"""cProfile of estimate_dev on the config-3 capture (GPU box): where the Python around the two native calls spends its time"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import spec_fir_taps, spec_ook_capture
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = spec_ook_capture(128, dev)
d_taps = torch.from_numpy(spec_fir_taps().view(np.float32).copy()).to(dev)
filt, noise = estimators.fir_filter_detect_noise_dev(pipe, iq, d_taps)
del iq
for _ in range(30):
    estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
