#!/usr/bin/env python3
"""debug: upload mode on small captures (one tile per chunk) under the chain knobs; which outputs differ from the oracle"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import urh_oracle as oracle
from test_stream_segments import _events_capture, _oracle_flat, _got, N
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
oracle.lib()
_lib.load().urhgpu_test_force_tiles_per_chunk(1)
p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
caps = [_events_capture(N, 70 + i) for i in range(2)]
refs = [_oracle_flat(oracle, c, p) for c in caps]
names = ("rows", "bits", "msg_off", "pauses", "pos", "pos_off")
def run(tag, tuning, upload=True, reps=2):
    pipe = DevicePipeline(0, pipelined=True, tuning=tuning)
    st = pipe.stream(N, p, want_qad=True, want_pos=True)
    out = []
    for rep in range(reps):
        for i, c in enumerate(caps):
            h = torch.from_numpy(c).pin_memory(); d = torch.zeros_like(h, device="cuda")
            r = st.push_upload(h, d) if upload else st.push(h.cuda())
            (r,) = st.flush()
            g = _got(r)
            bad = [nm for nm, a, b in zip(names, g, refs[i]) if not np.array_equal(a, b)]
            out.append("ok" if not bad else "BAD(" + ",".join(bad) + f" bits {len(g[1])}/{len(refs[i][1])})")
    print(f"{tag:50s} {out} {st.stats()['predicted_bytes']}", flush=True)
    st.close()
run("resident 8,0,3", {"stream_policy": 1, "stream_segments": 8, "stream_bits_segments": 3}, upload=False)
run("upload default (8 pieces)", {})
run("upload final_on_rows=0", {"stream_final_on_rows": 0})
run("upload fuse_gate=0", {"stream_fuse_gate": 0})
run("upload fuse_gate=0 final_on_rows=0", {"stream_fuse_gate": 0, "stream_final_on_rows": 0})
run("upload bits_segments=1", {"stream_bits_segments": 1})
run("upload bits_segments=8", {"stream_bits_segments": 8})
run("upload pieces=2", {"upload_pieces": 2})
run("upload pieces=3", {"upload_pieces": 3})
run("upload pieces=4 bits 2", {"upload_pieces": 4, "stream_bits_segments": 2})
