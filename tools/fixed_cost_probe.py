#!/usr/bin/env python3
"""Where a K-step measurement's FIXED cost goes (a K = 20 loop pays ~200 us once): host-side wall times of the phases of
bench.py's timed region -- the pushes, flush(), the final torch.cuda.synchronize() -- for K = 20, with and without the tail (urhgpu_test_tail_skip),
plus what an idle torch.cuda.synchronize() and a lone hipEventSynchronize cost.  python tools/fixed_cost_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dataclasses import replace
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True)
pipe.reserve(n, p)
lib = _lib.load()
st = pipe.stream(n, p, want_qad=True, want_pos=False)
def run(k):
    for _ in range(k): st.push(iq)
    st.flush()
for _ in range(6): run(30)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"idle torch.cuda.synchronize(): median {1e6*sorted(ts)[100]:.1f} us")
for mask in (0, 575, 0, 575):
    lib.urhgpu_test_tail_skip(mask)
    for K in (20, 80):
        rec = []
        for rep in range(7):
            run(30); torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_first = None
            for i in range(K):
                st.push(iq)
                if i == 0: t_first = time.perf_counter()
            t1 = time.perf_counter()
            st.flush()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rec.append((t3 - t0, t_first - t0, t1 - t0, t2 - t1, t3 - t2))
        rec.sort()
        tot, first, pushes, flush, sync = rec[len(rec) // 2]
        print(f"mask {mask:4d} K={K:3d}: {1e3*tot/K:.4f} ms/step; first push {1e6*first:6.1f} us, all pushes {1e6*pushes:8.1f}, flush {1e6*flush:7.1f}, final sync {1e6*sync:6.1f}")
lib.urhgpu_test_tail_skip(0)
st.close()
