#!/usr/bin/env python3
"""GFSK generator timing (developer tool; run on the GPU box): M messages of 2^20 samples each in one batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from urh_amd import signal_functions as sf

rng = np.random.default_rng(5)
for nmsg in (1, 16, 128):
    msgs = [rng.integers(0, 2, 10485).astype(np.uint8) for _ in range(nmsg)]
    for mod in ("FSK", "GFSK"):
        f = lambda: sf.modulate_messages_dev(msgs, 100, mod, [-20e3, 20e3], 1, 1.0, 40e3, 0.0, 1e6, [76] * nmsg, [0] * nmsg)
        out = f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{mod:5s} {nmsg:4d} messages, {out.shape[0]} samples: {dt * 1e3:8.2f} ms  {out.shape[0] / dt / 1e6:10.1f} Msamples/s")
