"""IQ->bits throughput per sample type (2^27 samples, 2-FSK): developer probe (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import iq_array
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, torch.device("cuda", 0), seed=1234)
n = iq.shape[0]
for dt in (np.float32, np.int16, np.int8, np.uint8):
    x = iq if dt == np.float32 else iq_array.convert_to((iq * 0.6).contiguous(), dt)
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    for _ in range(120): r = pipe.iq_to_bits(x, p, want_qad=True)      # 3 would do for the caches; ~100 passes bring the clocks up (tools/ramp_probe.py)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = pipe.iq_to_bits(x, p, want_qad=True)
    torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / 20
    bps = x.element_size() * 2 + 4
    note = "" if dt != np.uint8 else "  [uint8: afp_demod takes unsigned samples as they are (the reference's IQArray makes .cu8 captures signed first): the +128 offset is part of THIS signal -- other rows / bits, not comparable with the lines above]"
    print(f"{np.dtype(dt).name:8s} {dt_s * 1e3:.3f} ms/step  {n / dt_s / 1e9:.0f} Gsamples/s  {n * bps / dt_s / 1e12:.2f} TB/s algorithmic ({bps} B/sample)  counts {r.host_counts()[:3]}{note}")
