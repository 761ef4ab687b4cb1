#!/usr/bin/env python3
"""IQ->bits step time against the FSK deviation (developer tool, GPU box): the hot kernel's fast path covers phase steps below
atan(0.4375) = 0.412 rad per sample; beyond it every row takes the general atan2f."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math, torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
for dev_hz in ([float(a) * 1e3 for a in sys.argv[1:]] or (20e3, 50e3, 60e3, 70e3, 100e3, 200e3)):
    iq, _ = fsk_capture(128, torch.device("cuda", 0), seed=1234, deviation_hz=dev_hz)
    if os.environ.get("URH_DEV_DTYPE") == "int8": iq = (iq * 64.0).round().clamp(-127, 127).to(torch.int8).contiguous()      # (an int8 capture of the same signal)
    for _ in range(120): r = pipe.iq_to_bits(iq, p, want_qad=True)      # ~100 passes bring the clocks up (tools/ramp_probe.py)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = pipe.iq_to_bits(iq, p, want_qad=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"deviation {dev_hz / 1e3:5.0f} kHz  step {2 * math.pi * dev_hz / 1e6:.3f} rad/sample  {dt * 1e3:.3f} ms/step  {iq.shape[0] / dt / 1e9:.0f} Gsamples/s")
    del iq
