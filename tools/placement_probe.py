#!/usr/bin/env python3
"""Does the placement of the qad output buffer relative to the IQ input matter for the hot kernel?  (DESIGN.md section 7 noted 7 % between
two contexts with their own buffers.)  One 1 GiB capture, one arena for qad; the hot kernel's dispatch-attached timing for a sweep of
byte offsets of the qad buffer inside the arena.  usage: python tools/placement_probe.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = DevicePipeline(0)
pipe.reserve(n, p)
slack = 64 << 20
arena = torch.empty(n * 4 + slack, dtype=torch.uint8, device=dev)
print("iq at 0x%x, arena at 0x%x" % (iq.data_ptr(), arena.data_ptr()))
res = pipe.iq_to_bits(iq, p, want_qad=True)          # allocate the other buffers
lib = _lib.load()


def run(qad_ptr):
    cp = p.to_c("float32")
    cap_rows, cap_bits, cap_msg, cap_pos = pipe.capacities(n, p)
    o = _lib.Outputs()
    o.qad = qad_ptr
    o.rows = res.rows_buf.data_ptr(); o.cap_rows = cap_rows
    o.bits = res.bits_buf.data_ptr(); o.cap_bits = cap_bits
    o.msg_off = res.msg_off_buf.data_ptr(); o.pauses = res.pauses_buf.data_ptr(); o.cap_msg = cap_msg
    o.pos = res.pos_buf.data_ptr(); o.cap_pos = cap_pos; o.pos_off = res.pos_off_buf.data_ptr(); o.counts = res.counts.data_ptr()
    pipe.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(60):
        _lib.check(lib.urhgpu_iq_to_bits_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
    torch.cuda.synchronize()
    pipe.ctx.profile_begin(reps)
    for _ in range(reps):
        _lib.check(lib.urhgpu_iq_to_bits_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
    ms = pipe.ctx.profile_end()
    ms.sort()
    return ms[len(ms) // 2], ms[0]


base = arena.data_ptr()
offs = [0, 256, 1024, 4096, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20,
        (1 << 20) + 4096, (2 << 20) + 65536, (8 << 20) + (256 << 10)]
out = []
for off in offs:
    med, mn = run(base + off)
    out.append((off, med, mn))
    print("qad offset %10d  (iq-qad delta mod 2MiB %8d)  kernel median %.4f ms  min %.4f ms" % (off, (base + off - iq.data_ptr()) % (2 << 20), med, mn), flush=True)
meds = [m for _, m, _ in out]
print("spread: min %.4f max %.4f (%.1f %%)" % (min(meds), max(meds), (max(meds) / min(meds) - 1) * 100))
