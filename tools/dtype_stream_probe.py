"""Pipelined steps through the capture stream per sample type (what bench.py's `variants` time): python tools/dtype_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dataclasses import replace
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import spec_fsk_capture
dev = torch.device("cuda", 0)
iq, _ = spec_fsk_capture(128, dev, first_segment=0, sps=100)
n = iq.shape[0]
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
pipe = DevicePipeline(0, pipelined=True)
pipe.reserve(n, p)
print(f"# {torch.cuda.get_device_name(0)}; library {os.environ.get('URHGPU_LIB', 'urh_amd/liburhgpu.so')}")
for name, tdt, ndt, scale, bps in (("float32", torch.float32, np.float32, 1.0, 12), ("int16", torch.int16, np.int16, 8192.0, 8), ("int8", torch.int8, np.int8, 64.0, 6)):
    x = iq if ndt is np.float32 else (iq * scale).round().to(tdt).contiguous()
    st = pipe.stream(n, p, want_qad=True, want_pos=False, dtype=ndt)
    def run(k):
        for _ in range(k):
            st.push(x)
        st.flush()
    for _ in range(4):
        run(30)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    print(f"{name:8s} {ms:.4f} ms per pipelined step = {n * bps / (ms * 1e-3) / 8e12:.3f} of 8 TB/s at {bps} B/sample")
    st.close()
