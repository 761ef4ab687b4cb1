#!/usr/bin/env python3
"""Kernel-level view of the estimator passes at full size (run under rocprofv3 --kernel-trace --stats):
  config 3 stages: 1 GiB OOK -> FIR + noise statistics -> estimate (5 times);  detect_center on a 1 GiB demodulated PSK capture (5 times).
Prints wall times; the per-kernel averages come from the profiler's kernel stats."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from urh_amd import estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import spec_fir_taps, spec_ook_capture, spec_psk_capture


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return r, sorted(ts)[len(ts) // 2]


def main():
    import numpy as np
    dev = torch.device("cuda", 0)
    pipe = DevicePipeline(0)
    seg = int(os.environ.get("SEGMENTS", "128"))
    out = {}
    if "--no-ook" not in sys.argv:
        iq, _ = spec_ook_capture(seg, dev)
        taps = spec_fir_taps()
        d_taps = torch.from_numpy(taps.view(np.float32).copy()).to(dev)
        (filt, noise), out["fir_noise_ms"] = timed(lambda: estimators.fir_filter_detect_noise_dev(pipe, iq, d_taps))
        del iq
        for _ in range(10):
            estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK")
        est, out["estimate_ms"] = timed(lambda: estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK"))
        st = {}
        estimators.estimate_dev(pipe, filt, noise=noise, modulation="OOK", timings=st)
        out["estimate_stages_ms"] = st
        out["estimate"] = {k: (float(v) if not isinstance(v, str) else v) for k, v in (est or {}).items()}
        del filt
    if "--no-psk" not in sys.argv:
        from urh_amd.signal import Signal
        iq, _ = spec_psk_capture(seg, dev)
        sig = Signal(iq, modulation="PSK", pipe=pipe)
        del iq
        sig.bits_per_symbol = 2
        sig.noise_threshold = 0.2
        sig.center_spacing = 1.5
        sig.costas_loop_bandwidth = 0.1
        qad = sig.qad
        for _ in range(10):
            estimators.detect_center_dev(pipe, qad)
        c, out["detect_center_ms"] = timed(lambda: estimators.detect_center_dev(pipe, qad))
        out["center"] = float(c) if c is not None else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
