"""Band-pass on a 2^27-sample capture against the filter bandwidth (developer probe, GPU box): direct form (<= 95 taps) / LDS-FFT overlap-save"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import filter as uf
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0)
iq, _ = fsk_capture(int(sys.argv[1]) if len(sys.argv) > 1 else 128, torch.device("cuda", 0), seed=1)
for bw in (0.08, 0.045, 0.04, 0.01, 0.004, 0.001):
    m = len(uf.bandpass_taps(0.1, 0.2, bw))
    for _ in range(2): y = uf.apply_bandpass_filter_dev(pipe, iq, 0.1, 0.2, bw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): y = uf.apply_bandpass_filter_dev(pipe, iq, 0.1, 0.2, bw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"bw {bw:6.3f}  {m:5d} taps  {dt * 1e3:8.2f} ms  {iq.shape[0] / dt / 1e9:6.1f} Gsamples/s")
