"""What each kernel of the tail costs a pipelined step over the 10-samples-per-symbol capture (bench.py's variants.sps10: 6.7 M rows per GiB):
K = 40 steps with urhgpu_test_tail_skip(mask) leaving kernels out (bit 0 k_resolve_one, 1 k_emit_rows_tiles, 2 k_tile_scan, 3 the group scan,
4 k_expand_tiles, 5 the pack kernel; every pass processes the same capture, so the buffers still hold the previous pass's outputs)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from urh_amd import _lib
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
sps = int(os.environ.get("URH_SPS", "10"))
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=sps)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1 if sps < 50 else 5, sps, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True, tuning={k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")})
pipe.reserve(n, p)
st = pipe.stream(n, p, want_qad=os.environ.get("URH_NO_QAD") is None, want_pos=False, dtype=np.float32)      # URH_NO_QAD=1: the bits-only pass (the hot kernel writes nothing but its records)
lib = _lib.load()
def run(k):
    for _ in range(k): st.push(iq)
    st.flush()
for _ in range(6): run(20)
print(f"# {sps} samples per symbol, K = 40, ms per step (min of 3)")
masks = [int(a) for a in sys.argv[1:]]
for mask, what in ([(m, "mask from the command line (bit 9: no staged copies, 11: expansion without byte stores, 6: rows without blob stores)") for m in masks] if masks else []) + list(() if masks else ((0, "product"), (8, "without the group scan"), (16, "without k_expand_tiles"), (24, "without both"), (32, "without the pack kernel"),
                   (2, "without k_emit_rows_tiles"), (4, "without k_tile_scan"), (62, "resolve only"), (63, "no tail"), (0, "product"))):
    lib.urhgpu_test_tail_skip(mask)
    run(20); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"mask {mask:2d}  {min(ts):.4f}  {what}", flush=True)
lib.urhgpu_test_tail_skip(0)
st.close()
