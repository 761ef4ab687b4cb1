#!/usr/bin/env python3
"""Why is the hot kernel slower inside pipelined passes (0.281 ms) than on its own (0.267 ms on the same 224 CUs)?  Pipelined passes
with the full tail (rows, bits, positions), with rows only (two small tail kernels) and one after the other: hot kernel duration from
the dispatch-attached events, and the step time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
lib = _lib.load()


def run(pipelined, rows_only, reps=100):
    pipe = DevicePipeline(0, pipelined=pipelined)
    pipe.reserve(n, p)
    res = pipe.iq_to_bits(iq, p, want_qad=True)
    pipe.ctx.join(); torch.cuda.synchronize()
    cp = p.to_c("float32")
    cap_rows, cap_bits, cap_msg, cap_pos = pipe.capacities(n, p)
    o = _lib.Outputs()
    o.qad = res.qad.data_ptr()
    o.rows = res.rows_buf.data_ptr(); o.cap_rows = cap_rows
    o.counts = res.counts.data_ptr()
    if not rows_only:
        o.bits = res.bits_buf.data_ptr(); o.cap_bits = cap_bits
        o.msg_off = res.msg_off_buf.data_ptr(); o.pauses = res.pauses_buf.data_ptr(); o.cap_msg = cap_msg
        o.pos = res.pos_buf.data_ptr(); o.cap_pos = cap_pos; o.pos_off = res.pos_off_buf.data_ptr()
    pipe.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def steps(k):
        for _ in range(k):
            _lib.check(lib.urhgpu_iq_to_bits_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
        pipe.ctx.join(); torch.cuda.synchronize()
    steps(150)
    pipe.ctx.profile_begin(reps)
    t0 = time.perf_counter()
    steps(reps)
    dt = (time.perf_counter() - t0) / reps * 1e3
    ms = sorted(pipe.ctx.profile_end())
    if pipelined:
        pipe.ctx.set_pipelined(False)
    return round(dt, 4), round(ms[len(ms) // 2], 4), round(ms[0], 4)


for rep in range(2):
    for pipelined, rows_only, name in ((True, False, "pipelined, full tail"), (True, True, "pipelined, rows only"), (False, False, "one after the other, full tail"),
                                       (False, True, "one after the other, rows only")):
        step, med, mn = run(pipelined, rows_only)
        print(f"{name:34s} step {step} ms   hot kernel median {med} min {mn}", flush=True)
