#!/usr/bin/env python3
"""Per-stage timings of the secondary kernels at BASELINE sizes (developer tool; run on the GPU box):
FIR (64 taps), magnitude chunk statistics, qad-only demodulation, Costas loop, estimator passes."""
import ctypes as C
import json
import sys
import time

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from urh_amd import _lib, estimators
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture

dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
lib, h = _lib.load(), pipe.ctx.handle
segs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iq, _ = fsk_capture(segs, dev, seed=1)
n = iq.shape[0]
out = {"n": n}


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
taps = torch.from_numpy((np.random.default_rng(0).standard_normal((64, 2)) * 0.1).astype(np.float32)).to(dev)
fout = torch.empty_like(iq)
t = timed(lambda: _lib.check(lib.urhgpu_fir_filter_dev(h, C.c_void_p(iq.data_ptr()), n, C.c_void_p(taps.data_ptr()), 64, None,
                                                        C.c_void_p(fout.data_ptr()))))
out["fir64"] = {"ms": t * 1e3, "Msamples/s": n / t / 1e6, "GB/s(16B/sample)": 16 * n / t / 1e9, "GFLOP/s(512/sample)": 512 * n / t / 1e9}
t = timed(lambda: estimators.detect_noise_level_dev(pipe, iq))
out["detect_noise_level"] = {"ms": t * 1e3, "GB/s(8B/sample)": 8 * n / t / 1e9}
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100)
t = timed(lambda: pipe.afp_demod(iq, p))
out["afp_demod_fsk_qad_only"] = {"ms": t * 1e3, "GB/s(12B/sample)": 12 * n / t / 1e9}
pa = DemodParams("ASK", 1, 0.02, 0.3, 1.0, 5, 100)
t = timed(lambda: pipe.iq_to_bits(iq, pa, want_qad=True))
out["iq_to_bits_ask"] = {"ms": t * 1e3, "Msamples/s": n / t / 1e6}
t = timed(lambda: estimators.segment_messages_dev(pipe, iq, 0.3), reps=3)
out["segment_messages"] = {"ms": t * 1e3, "GB/s(8B/sample)": 8 * n / t / 1e9}
qad = pipe.afp_demod(iq, p)
t = timed(lambda: estimators.detect_center_dev(pipe, qad), reps=3)
out["detect_center"] = {"ms": t * 1e3, "GB/s(4B/sample)": 4 * n / t / 1e9}
t = timed(lambda: estimators.get_plateau_lengths_dev(pipe, qad, 0.0), reps=3)
out["get_plateau_lengths"] = {"ms": t * 1e3}
import math
m = n
k = torch.arange(m, device=dev, dtype=torch.float64)
sym = torch.randint(0, 4, (m // 100 + 1,), device=dev).repeat_interleave(100)[:m]
ph = (sym.to(torch.float64) * (math.pi / 2) - 3 * math.pi / 4) + 2 * math.pi * 0.04 * k
psk = torch.stack([torch.cos(ph), torch.sin(ph)], 1).to(torch.float32) + 0.07 * torch.randn((m, 2), device=dev)
del k, sym, ph
pp = DemodParams("PSK", 2, 0.2, 0.0, 1.5, 5, 100)
t = timed(lambda: pipe.afp_demod(psk, pp), reps=2)
out["costas_order4"] = {"samples": m, "ms": t * 1e3, "Msamples/s": m / t / 1e6, "chunks(map,ckpt,serial)": pipe.ctx.costas_stats()}
del psk
# ---- the steps either side of the path ----
from urh_amd import signal_functions as sf
from urh_amd.path_creator import create_path_arrays
from urh_amd.spectrogram import Spectrogram
rng = np.random.default_rng(5)
msgs = [rng.integers(0, 2, 10485).astype(np.uint8) for _ in range(segs)]
t = timed(lambda: sf.modulate_messages_dev(msgs, 100, "FSK", [-20e3, 20e3], 1, 1.0, 40e3, 0.0, 1e6, [76] * segs, [0] * segs), reps=3)
out["modulate_fsk_%d_messages" % segs] = {"ms": t * 1e3, "Msamples/s": n / t / 1e6, "GB/s(8B/sample written)": 8 * n / t / 1e9,
                                           "note": "includes the host-side staging of the bits"}
t = timed(lambda: create_path_arrays(qad, 0, n), reps=3)
out["path_minmax_qad"] = {"ms": t * 1e3, "GB/s(4B/sample)": 4 * n / t / 1e9}
spec = Spectrogram(iq[: n // 4])
t = timed(lambda: spec.calculate_spectrogram(device=True), reps=3)
out["spectrogram_db_1024_quarter"] = {"samples": n // 4, "ms": t * 1e3, "Msamples/s": n / 4 / t / 1e6,
                                       "GB/s(16B read + 8B written per sample)": 24 * (n // 4) / t / 1e9}
print(json.dumps(out, indent=1))
