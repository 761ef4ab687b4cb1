import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import estimators
from urh_amd.pipeline import DemodParams, DevicePipeline
from urh_amd.synth import fsk_capture
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = fsk_capture(128, dev, seed=1)
pa = DemodParams("ASK", 1, 0.02, 0.3, 1.0, 5, 100)
for _ in range(3):
    r = pipe.iq_to_bits(iq, pa, want_qad=True)
torch.cuda.synchronize()
print(r.host_counts())
p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100)
qad = pipe.afp_demod(iq, p)
for _ in range(2):
    c = estimators.detect_center_dev(pipe, qad)
    pl = estimators.get_plateau_lengths_dev(pipe, qad, 0.0)
print(c, len(pl))
