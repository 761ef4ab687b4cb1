#!/usr/bin/env python3
"""Round 4: one capture that starts on the host.  Bare pinned H2D copy of 1 GiB against urhgpu_stream_push_upload (pieces copied and
demodulated as they land, results in pinned host memory) for a few piece counts; and the resident single capture for the chain knobs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture

dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pinned = torch.empty(iq.shape, dtype=iq.dtype, pin_memory=True)
pinned.copy_(iq)
dst = torch.empty_like(iq)
torch.cuda.synchronize()
bare = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dst.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize(); bare.append((time.perf_counter() - t0) * 1e3)
print(f"bare pinned H2D of {n * 8 / 2**30:.2f} GiB: min {min(bare):.3f} ms ({n * 8 / min(bare) / 1e6:.1f} GB/s), all {[round(x, 3) for x in bare]}", flush=True)
for pieces in [int(x) for x in (sys.argv[1:] or ["8", "4", "16", "2"])]:
    tun = {"upload_pieces": pieces}
    tun.update(dict((k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("UP_TUNE", "").split(",") if kv)))
    pipe = DevicePipeline(0, pipelined=True, tuning=tun)
    pipe.reserve(n, p)
    st = pipe.stream(n, p, want_qad=True, want_pos=False)
    st.push(iq); st.flush()
    t = []
    for _ in range(6):
        dst.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st.push_upload(pinned, dst); r = st.flush()
        t.append((time.perf_counter() - t0) * 1e3)
    ok = bool(torch.equal(dst, iq))
    print(f"{os.environ.get('UP_TUNE', '')} upload in {pieces:2d} pieces: min {min(t):.3f} ms = bare + {min(t) - min(bare):+.3f} ms ({min(t) / min(bare):.4f} x), all {[round(x, 3) for x in t]}; "
          f"rows {r[-1].n_rows} bits {r[-1].n_bits}; device copy equal {ok}; {st.stats()}", flush=True)
    st.close(); del st, pipe
for tun in (() if os.environ.get("UP_ONLY") else ({}, {"stream_policy": 2})):
    pipe = DevicePipeline(0, pipelined=True, tuning=tun)
    pipe.reserve(n, p)
    st = pipe.stream(n, p, want_qad=True, want_pos=False)
    for _ in range(3):
        st.push(iq); st.flush()
    for _ in range(150):
        st.push(iq)
    st.flush(); torch.cuda.synchronize()
    one = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.push(iq); st.flush()
        one.append((time.perf_counter() - t0) * 1e3)
    one.sort()
    print(f"resident single capture {tun}: min {one[0]:.4f} median {one[6]:.4f} ms", flush=True)
    st.close(); del st, pipe
