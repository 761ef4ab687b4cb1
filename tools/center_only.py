"""detect_center on a 1 GiB capture's demodulated signal: target of rocprofv3 --kernel-trace --stats"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import estimators
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0)
iq, _ = fsk_capture(int(sys.argv[1]) if len(sys.argv) > 1 else 128, torch.device("cuda", 0), seed=1)
qad = pipe.afp_demod(iq, DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100))
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = estimators.detect_center_dev(pipe, qad)
    torch.cuda.synchronize(); print("center", c, "ms", (time.perf_counter() - t0) * 1e3)
