#!/usr/bin/env python3
"""Where does estimate_dev's wall time go: inside the three native calls (kernels + their host part) or in the Python between them?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import _lib, estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import spec_fir_taps, spec_ook_capture
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = spec_ook_capture(128, dev)
d_taps = torch.from_numpy(spec_fir_taps().view(np.float32).copy()).to(dev)
filt, noise = estimators.fir_filter_detect_noise_dev(pipe, iq, d_taps)
del iq
lib = _lib.load()
acc = {}
class Timed:
    def __init__(self, lib): self.__dict__["_l"] = lib
    def __getattr__(self, name):
        f = getattr(self._l, name)
        def w(*a):
            t = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
        return w
timed = Timed(lib)
orig = _lib.load
_lib.load = lambda: timed
for _ in range(30):
    estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK")
import gc; gc.collect(); gc.disable()
acc.clear()
K = 50
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK")
torch.cuda.synchronize(); total = (time.perf_counter() - t0) / K
print(json.dumps({"estimate_ms": round(total * 1e3, 4), "native_calls_ms": {k: round(v / K * 1e3, 4) for k, v in acc.items()},
                  "python_between_ms": round((total - sum(acc.values()) / K) * 1e3, 4)}))
