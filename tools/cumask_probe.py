#!/usr/bin/env python3
"""How many CUs should the hot kernel run on?  (Round 3: with 16 / 32 CUs masked out for an experiment it ran 4 / 7 % FASTER.)
The hot kernel (k_demod_runs_bp, 1 GiB complex64 2-FSK, passes one after the other) and the pure copy of its shape on streams whose CU
mask removes r CUs per XCD, r = 0 .. 16.  The removed set is balanced over the XCDs whichever way mask bits map to them."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = DevicePipeline(0)
pipe.reserve(n, p)
res = pipe.iq_to_bits(iq, p, want_qad=True)
torch.cuda.synchronize()
lib = _lib.load()
ncu = pipe.ctx.info()["compute_units"]
words = (ncu + 31) // 32
out_copy = torch.empty(n, dtype=torch.float32, device=dev)


def mask_for(r):
    """remove r CUs per XCD: class c = (i % 8 - i // 32) mod 8, k = (i // 8) % 4; removed iff (c, k) is among the first r pairs"""
    m = [0] * words
    for i in range(ncu):
        c, k = (i % 8 - i // 32) % 8, (i // 8) % 4
        removed = (k * 8 + c) < r * 2 if False else (c * 4 + k) < r
        if not removed:
            m[i // 32] |= 1 << (i % 32)
    return (C.c_uint32 * words)(*m), sum(bin(x).count("1") for x in m)


def run(stream_ptr, reps=30):
    cp = p.to_c("float32")
    cap_rows, cap_bits, cap_msg, cap_pos = pipe.capacities(n, p)
    o = _lib.Outputs()
    o.qad = res.qad.data_ptr()
    o.rows = res.rows_buf.data_ptr(); o.cap_rows = cap_rows
    o.bits = res.bits_buf.data_ptr(); o.cap_bits = cap_bits
    o.msg_off = res.msg_off_buf.data_ptr(); o.pauses = res.pauses_buf.data_ptr(); o.cap_msg = cap_msg
    o.pos = res.pos_buf.data_ptr(); o.cap_pos = cap_pos; o.pos_off = res.pos_off_buf.data_ptr(); o.counts = res.counts.data_ptr()
    pipe.ctx.set_stream(stream_ptr)
    for _ in range(80):
        _lib.check(lib.urhgpu_iq_to_bits_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
    pipe.ctx.sync()
    pipe.ctx.profile_begin(reps)
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.urhgpu_iq_to_bits_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
    ms = pipe.ctx.profile_end()
    step = (time.perf_counter() - t0) / reps * 1e3
    ms.sort()
    cms = C.c_float(0.0)
    _lib.check(lib.urhgpu_bench_copy_ceiling_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), C.c_void_p(out_copy.data_ptr()), n, 0, 40, C.byref(cms)))
    return ms[len(ms) // 2], ms[0], step, cms.value


for r in (0, 1, 2, 3, 4, 6, 8, 12, 16):
    mask, used = mask_for(r)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), mask)
    assert rc == 0, rc
    med, mn, step, copy_ms = run(s.value)
    print("CUs used %3d (removed %2d per XCD): hot kernel median %.4f min %.4f ms = %.0f GB/s | unpipelined step %.4f ms | shape copy %.4f ms = %.0f GB/s"
          % (used, r, med, mn, n * 12 / med / 1e6, step, copy_ms, n * 12 / copy_ms / 1e6), flush=True)
    pipe.ctx.sync()
    hip.hipStreamDestroy(s)
