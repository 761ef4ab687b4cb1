"""IQ->bits per sample type (2^27 samples, 2-FSK): un-pipelined step and the hot kernel's own duration (HIP events on its
dispatch) for complex64 / int16 / int8 captures of the same signal, against the bytes each moves.  Developer probe (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import iq_array
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, torch.device("cuda", 0), seed=1234)
n = iq.shape[0]
print(f"# {torch.cuda.get_device_name(0)}; library {os.environ.get('URHGPU_LIB', 'urh_amd/liburhgpu.so')}")
for dt in (np.float32, np.int16, np.int8):
    x = iq if dt == np.float32 else iq_array.convert_to((iq * 0.6).contiguous(), dt)
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    for _ in range(120): r = pipe.iq_to_bits(x, p, want_qad=True)      # ~100 passes bring the clocks up
    torch.cuda.synchronize()
    pipe.ctx.profile_begin(20)
    t0 = time.perf_counter()
    for _ in range(20): r = pipe.iq_to_bits(x, p, want_qad=True)
    torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / 20
    k = pipe.ctx.profile_end()
    k_ms = sum(k) / max(len(k), 1)
    bps = x.element_size() * 2 + 4
    print(f"{np.dtype(dt).name:8s} step {dt_s * 1e3:.3f} ms  hot kernel {k_ms:.4f} ms = {n * bps / (k_ms * 1e-3) / 1e12 if k_ms else 0:.2f} TB/s = "
          f"{n * bps / (k_ms * 1e-3) / 8e12 if k_ms else 0:.3f} of 8 TB/s ({bps} B/sample)  counts {r.host_counts()[:3]}")
