#!/usr/bin/env python3
"""Anatomy of the boundary between two consecutive hot kernels (profiles/r05_boundary_anatomy.txt).

`urhgpu_test_hot_probe` launches the hot kernel alone, back to back, in its STAMPS instantiation: wavefront 0 of every workgroup leaves
three s_memrealtime stamps (100 MHz: 10 ns) -- entry, streaming phase over, ChunkInfo written -- and its hardware ids.  From the last
launches (warm clocks) this prints, per variant (stream, events attached to the dispatch, graded tail):
  * the period (first wavefront's entry of launch j + 1 minus that of launch j), the wave-level duration (first entry -> last end) and
    the wave-level gap (last end of launch j -> first entry of launch j + 1); with timing events (mode 3) the dispatches' own
    durations and gaps next to them: what the dispatch adds around the waves;
  * the occupancy curve (workgroups alive) over the first and the last 32 us of a launch: ramp and drain;
  * the life of a workgroup (entry -> end; streaming part; run phase) and where a chunk's predecessors stand when it ends (what a
    chained look-back inside the hot kernel would have to wait for).

    python tools/boundary_probe.py [--segments 128] [--launches 150] [--keep 12] [--variants A,B,...]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNK = np.dtype([("pend_pos", "<i8"), ("lead", "<i8"), ("start", "<i8"), ("len", "<i8"), ("last_pos", "<i8"), ("cnt", "<i4"),
                  ("first_state", "<u2"), ("last_state", "<u2"), ("pend_state", "<u2"), ("init_state", "<u2"),
                  ("t_entry", "<u4"), ("hw", "<u4"), ("t_streamed", "<u4"), ("t_end", "<u4"), ("_pad", "<u4")])
assert CHUNK.itemsize == 72

VARIANTS = {                      # name: (stream_kind, event_mode, graded, what)
    "A": (0, 0, 0, "caller's stream (256 CUs), plain launches"),
    "B": (1, 0, 0, "CU-masked hot stream (224 CUs), plain launches"),
    "C": (1, 1, 0, "masked, completion event on every dispatch (what pipelined passes do)"),
    "D": (1, 2, 0, "masked, completion event created with DisableSystemFence | ReleaseToDevice"),
    "E": (1, 3, 0, "masked, timing events on every dispatch (dispatch-level durations and gaps)"),
    "F": (1, 1, 392, "masked, completion event, graded tail: last 392 chunks (0.25 residency waves) as 16-row chunks"),
    "G": (1, 1, 784, "masked, completion event, graded tail: last 784 chunks (0.5)"),
    "H": (1, 1, 1568, "masked, completion event, graded tail: last 1568 chunks (1.0)"),
    "I": (1, 1, 2352, "masked, completion event, graded tail: last 2352 chunks (1.5)"),
    "J": (0, 1, 784, "256 CUs, completion event, graded tail: last 784 chunks"),
    "K": (1, 1, 0, "masked, completion event, a 20 us bubble between two hot kernels", 20),
    "L": (1, 1, 0, "masked, completion event, a 60 us bubble between two hot kernels", 60),
    "M": (0, 1, 0, "256 CUs, completion event, a 60 us bubble between two hot kernels", 60),
    "N": (1, 1, 0, "masked + company: 256 wavefronts of arithmetic for 200 us on the 32 CUs the mask leaves out", 0, 1),
    "O": (1, 1, 0, "masked + company: the same at s_setprio 3", 0, 2),
    "P": (1, 1, 0, "masked + company: 4096 workgroups x 4 wavefronts x 4 us of arithmetic at s_setprio 3, anywhere", 0, 3),
    "Q": (1, 1, 0, "masked + company: 256 wavefronts of dependent random loads for 200 us on the 32 CUs left out", 0, 4),
    "R": (1, 1, 0, "masked + company: six empty one-wavefront kernels in a row (kernel boundaries)", 0, 5),
}


def d32(a, b):
    """a - b on 32-bit wrapping stamps, as signed microseconds (10 ns units)"""
    return ((np.asarray(a, np.int64) - np.asarray(b, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)) / 100.0


def pct(x, qs=(50, 90, 99, 100)):
    x = np.asarray(x, np.float64)
    return " ".join(f"p{q}={np.percentile(x, q):.2f}" for q in qs)


def analyse(tab, graded=0, dur=None, gap=None, brief=False):
    """tab: (launches, chunks) structured array of consecutive launches, oldest first"""
    keep, nc = tab.shape
    ent, end, mid = tab["t_entry"], tab["t_end"], tab["t_streamed"]
    # reference points per launch (stamps wrap at 43 s: differences only)
    first = np.array([ent[j][np.argmin(d32(ent[j], ent[j][0]))] for j in range(keep)], np.int64)
    last = np.array([end[j][np.argmax(d32(end[j], ent[j][0]))] for j in range(keep)], np.int64)
    wdur = d32(last, first)
    print(f"   first wavefront's entry -> last wavefront's end {wdur.mean():8.2f} us  ({pct(wdur, (0, 50, 100))})")
    if keep > 1:
        period = d32(first[1:], first[:-1])
        wgap = d32(first[1:], last[:-1])
        print(f"   period           {period.mean():8.2f} us  ({pct(period, (0, 50, 100))})")
        print(f"   last end -> next launch's first entry          {wgap.mean():8.2f} us  ({pct(wgap, (0, 50, 100))})")
    if dur is not None:
        print(f"   dispatch-level (HIP events on the dispatch): duration {dur.mean():8.2f} us, gap {gap.mean():6.2f} us  -> dispatch begin..first entry + last end..dispatch end = "
              f"{dur.mean() - wdur.mean():.2f} us; of the gap {gap.mean():.2f} + that = {gap.mean() + dur.mean() - wdur.mean():.2f} us lie between the waves")
    j = max(keep - 2, 0)
    e0 = d32(ent[j], first[j]); e1 = d32(end[j], first[j]); m1 = d32(mid[j], first[j])
    life = e1 - e0
    print(f"   workgroup life   {pct(life, (1, 50, 90, 99, 100))} us; streaming part {pct(m1 - e0, (50, 99))}; run phase + ChunkInfo {pct(e1 - m1, (50, 99))}")
    if brief:
        return
    q = nc // 4
    print("   workgroup life by quarter of the launch (p50): " + " ".join(f"{np.percentile(life[k * q:(k + 1) * q], 50):.2f}" for k in range(4)))
    if graded:
        g0 = nc - 4 * graded
        print(f"     long chunks: {pct(life[:g0], (50, 99))}; short chunks: {pct(life[g0:], (50, 99))}")
    T1 = e1.max()
    ts = np.arange(0, 32.1, 2.0)
    ramp = [(int(((e0 <= t) & (e1 > t)).sum())) for t in ts]
    drain = [(int(((e0 <= T1 - t) & (e1 > T1 - t)).sum())) for t in ts]
    print("   workgroups alive at t = 0, 2, .. 32 us after the first entry: " + " ".join(map(str, ramp)))
    print("   workgroups alive at t = 0, 2, .. 32 us before the last end:   " + " ".join(map(str, drain)))
    if keep > 1 and j + 1 < keep:
        nxt0 = d32(ent[j + 1], first[j + 1])
        print(f"   entries of the NEXT launch's first residency wave: {pct(nxt0[:1500], (1, 50, 99))} us after its first entry")
    # in-order-ness and what a chained look-back would wait for
    order = np.argsort(e0, kind="stable")
    disp = np.abs(order - np.arange(nc))
    pm = np.maximum.accumulate(e1)
    wait = np.maximum(0.0, pm[:-1] - e1[1:])
    print(f"   start order vs chunk id: {float((np.diff(e0) < 0).mean()) * 100:.1f} % of chunks enter before their predecessor, displacement {pct(disp, (50, 99, 100))} chunks")
    print(f"   when chunk c ends, the LAST of its predecessors ends {pct(wait, (50, 90, 99, 100))} us later (mean {wait.mean():.2f}); "
          f"{float((wait > 0).mean()) * 100:.1f} % of chunks would wait at all")
    for grp in (16, 64, 256):
        ng = nc // grp
        gend = e1[: ng * grp].reshape(ng, grp).max(axis=1)      # when a group's last arriver arrives
        gw = np.maximum(0.0, np.maximum.accumulate(gend)[:-1] - gend[1:])
        print(f"   groups of {grp} chunks, the last arriver looks back over groups: it waits {pct(gw, (50, 90, 99, 100))} us (mean {gw.mean():.2f}); "
              f"the group is complete {np.mean(gend - e1[: ng * grp].reshape(ng, grp).mean(axis=1)):.2f} us after its average member")
    xcc = (tab["hw"][j] >> 16) & 0xF
    print("   workgroups per XCC id: " + " ".join(f"{k}:{int((xcc == k).sum())}" for k in range(8)) + f"; chunk id mod 8 == XCC id for {float((xcc == (np.arange(nc) % 8)).mean()) * 100:.1f} %")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--segments", type=int, default=128)
    ap.add_argument("--launches", type=int, default=150)
    ap.add_argument("--keep", type=int, default=12)
    ap.add_argument("--variants", default="A,B,C,E,F,G,H,I,J")
    args = ap.parse_args()
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    from urh_amd.synth import spec_fsk_capture
    dev = torch.device("cuda", 0)
    iq, _ = spec_fsk_capture(args.segments, dev, first_segment=0, sps=100)
    n = iq.shape[0]
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True)
    pipe.reserve(n, p)
    lib = _lib.load()
    cp = p.to_c(np.float32)
    qad = torch.empty(n, dtype=torch.float32, device=dev)
    cap_chunks = n // 8192 * 4 + 64
    out = torch.zeros(args.keep * cap_chunks * 72, dtype=torch.uint8, device=dev)
    print(f"# {torch.cuda.get_device_name(0)}; capture {n} samples; {args.launches} launches per variant, the last {args.keep} analysed")
    for name in args.variants.split(","):
        kind, mode, graded, what = VARIANTS[name][:4]
        bubble = VARIANTS[name][4] if len(VARIANTS[name]) > 4 else 0
        load = VARIANTS[name][5] if len(VARIANTS[name]) > 5 else 0
        nch = C.c_int64(0)
        dur = (C.c_float * args.launches)()
        gap = (C.c_float * args.launches)()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = lib.urhgpu_test_hot_probe(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.c_void_p(qad.data_ptr()), kind, mode, graded,
                                       args.launches, args.keep, C.c_void_p(out.data_ptr()), C.byref(nch), dur, gap, bubble, load)
        wall = time.perf_counter() - t0
        if st != 0:
            print(f"== {name}: {what}: status {st} ({lib.urhgpu_strerror(st).decode()}) {lib.urhgpu_last_hip_error().decode()}")
            continue
        nc = nch.value
        tab = out[: args.keep * nc * 72].cpu().numpy().view(CHUNK).reshape(args.keep, nc)
        print(f"\n== {name}: {what}   [{nc} workgroups, wall {wall / args.launches * 1e3:.4f} ms per launch incl. set-up]")
        analyse(tab, graded=graded, dur=np.array(dur[args.launches - args.keep:]) * 1e3 if mode == 3 else None,
                gap=np.array(gap[args.launches - args.keep: args.launches - 1]) * 1e3 if mode == 3 else None)


if __name__ == "__main__":
    main()
