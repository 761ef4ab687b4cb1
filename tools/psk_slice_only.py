"""configs[4]: slicing of the Costas-demodulated signal (qad -> pulse table -> bits), centre 0 and detected centre: target of rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd import estimators
from urh_amd.pipeline import DevicePipeline
from urh_amd.signal import Signal
from urh_amd.synth import spec_psk_capture
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
iq, _ = spec_psk_capture(int(sys.argv[1]) if len(sys.argv) > 1 else 128, dev)
sig = Signal(iq, modulation="PSK", pipe=pipe)
sig.bits_per_symbol = 2; sig.noise_threshold = 0.2; sig.center_spacing = 1.5; sig.costas_loop_bandwidth = 0.1
qad = sig.qad
for c in (0.0, float(estimators.detect_center_dev(pipe, qad))):
    sig.center = c
    for _ in range(3):
        sig._bits = None
        sig._digitize()
torch.cuda.synchronize()
