#!/usr/bin/env python3
"""The hot kernel's wave-level timeline INSIDE the product's loops (tools/boundary_probe.py measures the kernel alone): the stamped
instantiation runs in place of the product kernel (urhgpu_test_hot_stamps), the loops are bench.py's -- K pushes through the capture
stream (the headline), K pipelined device-only passes, K passes one after the other -- and the chunk tables of the last three passes
(pipelined loops) are read back and analysed as in the probe: duration first entry -> last end, gap to the next pass's first entry,
ramp and drain, the life of a workgroup by quarter of the launch.  What the tail beside the hot kernel costs it, and where.

    python tools/inrun_anatomy.py [--segments 128] [--steps 40]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from boundary_probe import CHUNK, analyse          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=128)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--no-stamps", action="store_true", help="the product kernel (no tables): wall-clock per step only")
    ap.add_argument("--skips", default="", help="comma-separated urhgpu_test_tail_skip masks: the headline loop once per mask, brief analysis (then nothing else)")
    args = ap.parse_args()
    import torch
    from dataclasses import replace
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    from urh_amd.synth import spec_fsk_capture
    dev = torch.device("cuda", 0)
    iq, _ = spec_fsk_capture(args.segments, dev, first_segment=0, sps=100)
    n = iq.shape[0]
    nc = n // 8192
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    lib = _lib.load()
    tuning = {k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")}
    pipe = DevicePipeline(0, pipelined=True, tuning=tuning)
    pipe.reserve(n, p)
    if not args.no_stamps:
        lib.urhgpu_test_hot_stamps(1)
    host = np.zeros(3 * nc, dtype=CHUNK)

    def tables(k=3):
        _lib.check(lib.urhgpu_test_fetch_chunk_tables(pipe.ctx.handle, host.ctypes.data_as(C.c_void_p), nc))
        t = host.reshape(3, nc)[:k][::-1].copy()            # oldest first
        return t

    def timed(fn, steps):
        for _ in range(4):                                   # clock ramp: ~100 passes
            fn(30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    print(f"# {torch.cuda.get_device_name(0)}; {n} samples; tuning {tuning}; stamps {'off' if args.no_stamps else 'on'}")
    if args.skips:
        names = ["k_resolve_one", "k_emit_rows_tiles", "k_tile_scan", "group scan", "k_expand_tiles", "k_pack_seg", "the row kernel's host stores", "the row kernel's int64 table", "(unused)", "the blob's copies", "(unused)", "the expansion's byte stores"]
        st = pipe.stream(n, replace(p, write_bit_sample_pos=False), want_qad=True, want_pos=False)

        def pushes(k):
            for _ in range(k):
                st.push(iq)
            st.flush()
        timed(pushes, args.steps)                            # every arena and host blob holds a full pass's outputs
        for mask in [int(x) for x in args.skips.split(",")]:
            lib.urhgpu_test_tail_skip(mask)
            ms = timed(pushes, args.steps)
            left = [nm for b, nm in enumerate(names) if mask >> b & 1]
            print(f"\n== headline loop without {', '.join(left) if left else 'nothing (the product)'} [mask {mask}]: {ms:.4f} ms per step (K = {args.steps})")
            if not args.no_stamps:
                analyse(tables(), brief=True)
        lib.urhgpu_test_tail_skip(0)
        st.close()
        return
    for want_pos, label in ((False, "capture stream, compact outputs to the host, no positions (the headline loop)"),
                            (True, "capture stream with bit_sample_pos produced and shipped")):
        st = pipe.stream(n, replace(p, write_bit_sample_pos=want_pos), want_qad=True, want_pos=want_pos)

        def pushes(k):
            for _ in range(k):
                st.push(iq)
            st.flush()
        ms = timed(pushes, args.steps)
        print(f"\n== {label}: {ms:.4f} ms per step (K = {args.steps})")
        if not args.no_stamps:
            analyse(tables())
        st.close()

    def device_steps(k):
        for _ in range(k):
            pipe.iq_to_bits(iq, p, want_qad=True)
        pipe.ctx.join()
    ms = timed(device_steps, args.steps)
    print(f"\n== pipelined device-only passes: {ms:.4f} ms per step")
    if not args.no_stamps:
        analyse(tables())
    pipe.ctx.set_pipelined(False)
    ms = timed(device_steps, args.steps)
    print(f"\n== passes one after the other (context not pipelined): {ms:.4f} ms per step")
    if not args.no_stamps:
        analyse(tables(1))
    lib.urhgpu_test_hot_stamps(0)


if __name__ == "__main__":
    main()
