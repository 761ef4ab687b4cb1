#!/usr/bin/env python3
"""urhgpu_fft_peak_dev (Signal.estimate_frequency's FFT) on 2^13 .. 2^26 samples: wall time per call incl. its one synchronisation, and
what that is against the passes' HBM traffic (four-step: two transposes + two row-FFT passes = 64 B per sample; one LDS transform: 16 B)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import _lib
from urh_amd.pipeline import DevicePipeline
pipe = DevicePipeline()
lib = _lib.load()
for k in (13, 16, 20, 24, 26):
    n = 1 << k
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    x = torch.stack([torch.cos(2 * np.pi * 0.123 * t), torch.sin(2 * np.pi * 0.123 * t)], 1).to(torch.float32).contiguous()
    peak = C.c_int64(0)
    pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.check(lib.urhgpu_fft_peak_dev(pipe.ctx.handle, C.c_void_p(x.data_ptr()), n, C.byref(peak)))
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        _lib.check(lib.urhgpu_fft_peak_dev(pipe.ctx.handle, C.c_void_p(x.data_ptr()), n, C.byref(peak)))
        ts.append(time.perf_counter() - t0)
    ms = sorted(ts)[len(ts) // 2] * 1e3
    traffic = n * (64 if k > 13 else 16) + n * 8
    print(f"n = 2^{k}: {ms:.3f} ms per call, peak bin {peak.value} (expected {round(0.123 * n)}), {traffic / ms / 1e6:.0f} GB/s of its {traffic / 1e6:.0f} MB")
