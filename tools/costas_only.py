"""Costas loop only (order 4, 1 GiB): target of rocprofv3 --kernel-trace --stats"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urh_amd.pipeline import DemodParams, DevicePipeline
dev = torch.device("cuda", 0)
pipe = DevicePipeline(0)
m = (int(sys.argv[1]) if len(sys.argv) > 1 else 128) << 20
k = torch.arange(m, device=dev, dtype=torch.float64)
sym = torch.randint(0, 4, (m // 100 + 1,), device=dev).repeat_interleave(100)[:m]
ph = (sym.to(torch.float64) * (math.pi / 2) - 3 * math.pi / 4) + 2 * math.pi * 0.04 * k
psk = torch.stack([torch.cos(ph), torch.sin(ph)], 1).to(torch.float32) + 0.07 * torch.randn((m, 2), device=dev)
del k, sym, ph
pp = DemodParams("PSK", 2, 0.2, 0.0, 1.5, 5, 100)
for _ in range(3):
    q = pipe.afp_demod(psk, pp)
torch.cuda.synchronize()
print(pipe.ctx.costas_stats())
