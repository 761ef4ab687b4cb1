"""FIR only (64 taps, 1 GiB): target of tools/profiles_round.sh"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from urh_amd import _lib
from urh_amd.pipeline import DevicePipeline
from urh_amd.synth import fsk_capture
pipe = DevicePipeline(0); lib, h = _lib.load(), pipe.ctx.handle
iq, _ = fsk_capture(int(sys.argv[1]) if len(sys.argv) > 1 else 128, torch.device("cuda", 0), seed=1)
n = iq.shape[0]
taps = torch.from_numpy((np.random.default_rng(0).standard_normal((64, 2)) * 0.1).astype(np.float32)).cuda()
out = torch.empty_like(iq)
pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(6):
    _lib.check(lib.urhgpu_fir_filter_dev(h, C.c_void_p(iq.data_ptr()), n, C.c_void_p(taps.data_ptr()), 64, None, C.c_void_p(out.data_ptr())))
torch.cuda.synchronize()
