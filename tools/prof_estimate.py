"""Run AutoInterpretation.estimate's device chain a few times on the configs[2] capture (for rocprofv3 --kernel-trace --stats)."""
import argparse

import numpy as np
import torch

from urh_amd import estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import spec_fir_taps, spec_ook_capture

ap = argparse.ArgumentParser()
ap.add_argument("--segments", type=int, default=128)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe = DevicePipeline(0)
iq, _ = spec_ook_capture(a.segments, dev)
taps = torch.from_numpy(spec_fir_taps().view(np.float32).copy()).to(dev)
filt, noise = estimators.fir_filter_detect_noise_dev(pipe, iq, taps)
for _ in range(a.reps):
    t = {}
    est = estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK", timings=t)
print(est, t)
