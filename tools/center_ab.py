import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from urh_amd import estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import spec_psk_capture
from urh_amd.signal import Signal
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = spec_psk_capture(128, dev)
sig = Signal(iq, modulation="PSK", pipe=pipe)
del iq
sig.bits_per_symbol = 2; sig.noise_threshold = 0.2; sig.center_spacing = 1.5; sig.costas_loop_bandwidth = 0.1
qad = sig.qad
for _ in range(10): estimators.detect_center_dev(pipe, qad)
ts = []
for _ in range(9):
    torch.cuda.synchronize(); t = time.perf_counter(); c = estimators.detect_center_dev(pipe, qad); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(json.dumps({"detect_center_ms": round(sorted(ts)[4], 4), "center": float(c)}))
