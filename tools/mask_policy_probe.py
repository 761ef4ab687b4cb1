#!/usr/bin/env python3
"""Which captures gain from the CU-masked hot stream of pipelined passes?  Device-only pipelined steps (2^27 samples, 2-FSK) per sample
type and for a wide deviation, with the hot kernel on 224 CUs (mask 4 per XCD) and on all 256 (no mask)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import iq_array
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
dev = torch.device("cuda", 0)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
cases = [("float32", np.float32, 20e3), ("float32 +-100 kHz", np.float32, 100e3), ("int16", np.int16, 20e3), ("int8", np.int8, 20e3), ("uint8", np.uint8, 20e3)]
for name, dt, dev_hz in cases:
    iq, _ = fsk_capture(128, dev, seed=1234, deviation_hz=dev_hz)
    x = iq if dt == np.float32 else iq_array.convert_to((iq * 0.6).contiguous(), dt)
    del iq
    out = []
    for removed in (4, 0, 4, 0):
        pipe = DevicePipeline(0, pipelined=True, tuning={"hot_cus_removed_per_xcd": removed})
        pipe.reserve(x.shape[0], p)
        for _ in range(150):
            pipe.iq_to_bits(x, p, want_qad=True)
        pipe.ctx.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            pipe.iq_to_bits(x, p, want_qad=True)
        pipe.ctx.join(); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 60 * 1e3, 4))
        pipe.ctx.set_pipelined(False)
        del pipe
    print(f"{name:20s} ms/step masked(224 CUs) / all 256 / masked / all: {out}", flush=True)
    del x
    torch.cuda.empty_cache()
