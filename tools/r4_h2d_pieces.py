#!/usr/bin/env python3
"""How long do the PIECES of a pinned H2D copy take?  Plain torch copies on one side stream, an event pair around every piece; no kernels."""
import sys, time, torch
n = 1 << 27
pinned = torch.empty((n, 2), dtype=torch.float32, pin_memory=True); pinned.normal_()
dst = torch.empty((n, 2), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
unit = 256 * 8192
for pieces in [int(x) for x in (sys.argv[1:] or ["2", "4", "8", "16"])]:
    blocks = n // unit
    rest = blocks - 1
    bounds = [0] + [(rest * k // (pieces - 1)) * unit + 8192 for k in range(1, pieces)] + [n]
    best = None
    for rep in range(3):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(pieces + 1)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(side):
            evs[0].record(side)
            for k in range(pieces):
                dst[bounds[k]:bounds[k + 1]].copy_(pinned[bounds[k]:bounds[k + 1]], non_blocking=True)
                evs[k + 1].record(side)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        per = [evs[k].elapsed_time(evs[k + 1]) for k in range(pieces)]
        if best is None or dt < best[0]:
            best = (dt, per)
    sizes = [(bounds[k + 1] - bounds[k]) * 8 / 2**20 for k in range(pieces)]
    print(f"{pieces:2d} pieces: total {best[0]:.3f} ms; per piece ms {[round(x, 2) for x in best[1]]}; GB/s {[round(s * 2**20 / 1e6 / max(t, 1e-3), 1) for s, t in zip(sizes, best[1])]}", flush=True)
