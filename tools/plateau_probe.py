#!/usr/bin/env python3
"""Where get_plateau_lengths_dev spends its time (developer tool; run on the GPU box)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import _lib, estimators
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture

dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, dev, seed=1)
qad = pipe.afp_demod(iq, DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100))
n = qad.shape[0]
w = min(n, n // 4 + (1 << 16))
cap = max(1 << 16, w // 16)
idx = torch.empty(cap, dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
lib, h = _lib.load(), pipe.ctx.handle
pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def k():
    _lib.check(lib.urhgpu_edges_le_dev(h, C.c_void_p(qad.data_ptr()), w, 0.0, C.c_void_p(idx.data_ptr()), cap, C.c_void_p(cnt.data_ptr())))
for name, fn in (("edges kernel", k), ("cnt.item", lambda: cnt.item()), ("idx.cpu", lambda: idx[:int(cnt.item())].cpu().numpy()),
                 ("whole", lambda: estimators.get_plateau_lengths_dev(pipe, qad, 0.0))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print("%-14s %8.3f ms" % (name, (time.perf_counter() - t0) / 5 * 1e3))
print("boundaries", int(cnt.item()), "window", w)
