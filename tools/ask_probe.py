#!/usr/bin/env python3
"""ASK-side timings of the hot kernel (developer tool; run on the GPU box): OOK IQ->bits, message segmentation, Signal.qad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import estimators
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture

dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, dev, seed=1)
n = iq.shape[0]


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


pa = DemodParams("ASK", 1, 0.02, 0.3, 1.0, 5, 100)
print("iq_to_bits ASK       %.4f ms" % timed(lambda: pipe.iq_to_bits(iq, pa, want_qad=True)))
print("afp_demod ASK        %.4f ms" % timed(lambda: pipe.afp_demod(iq, pa)))
print("segment_messages     %.4f ms" % timed(lambda: estimators.segment_messages_dev(pipe, iq, 0.3), 5))
pf = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100)
print("iq_to_bits FSK       %.4f ms" % timed(lambda: pipe.iq_to_bits(iq, pf, want_qad=True)))
# modulation order 4 on the same capture (three thresholds): bit-plane kernel vs the state-byte kernel
from urh_amd import _lib
p4 = DemodParams("FSK", 2, 0.0, 0.0, 0.4, 5, 100)
for force in (0, 1):
    _lib.load().urhgpu_test_force_state_bytes(force)
    print("iq_to_bits FSK order 4, %s  %.4f ms" % ("state bytes" if force else "bit planes ", timed(lambda: pipe.iq_to_bits(iq, p4, want_qad=True))))
_lib.load().urhgpu_test_force_state_bytes(0)
