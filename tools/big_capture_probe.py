import sys, time
sys.path.insert(0, ".")
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
dev = torch.device("cuda", 0)
iq, tx = fsk_capture(512, dev, seed=1234)
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
pipe = DevicePipeline(0)
r = pipe.iq_to_bits(iq, p, want_qad=True)
c = r.host_counts(); r.check_capacity()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = pipe.iq_to_bits(iq, p, want_qad=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("n", iq.shape[0], "counts", c, "ms", dt * 1e3, "Gsamples/s", iq.shape[0] / dt / 1e9)
# first quarter must equal the 1 GiB result (same seeds: segments 0..127), up to the last row
r1 = pipe.iq_to_bits(iq[: 128 << 20], p, want_qad=True)
a = r.ppseq(); b = r1.ppseq()
print("prefix rows equal:", bool((a[: len(b) - 2] == b[: len(b) - 2]).all()), len(a), len(b))
