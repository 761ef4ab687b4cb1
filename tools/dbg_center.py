import sys, ctypes as C
sys.path[:0] = ['.', 'oracle', 'tests']
import numpy as np, torch
from conftest import load_golden
from urh_amd import _lib
from urh_amd.pipeline import DevicePipeline
pipe = DevicePipeline()
lib, h = _lib.load(), pipe.ctx.handle
q = load_golden("fsk")["qad"]
x = torch.from_numpy(q).cuda()
n = len(q)
kept = torch.empty(n, dtype=torch.float32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.urhgpu_compact_gt_dev(h, C.c_void_p(x.data_ptr()), n, -4.0, C.c_void_p(kept.data_ptr()), C.c_void_p(cnt.data_ptr())))
k = int(cnt.item())
ref = q[q > -4]
print("k", k, len(ref), np.array_equal(kept[:k].cpu().numpy(), ref))
a, b = int(0.05 * k), int(0.95 * k)
r = kept[a:b]; rr = ref[a:b]; m = b - a
mm = torch.empty(2, dtype=torch.float32, device="cuda")
_lib.check(lib.urhgpu_minmax_f32_dev(h, C.c_void_p(r.data_ptr()), m, C.c_void_p(mm.data_ptr())))
print("minmax", mm.cpu().numpy(), rr.min(), rr.max())
s = C.c_float(0)
_lib.check(lib.urhgpu_pairwise_sum_f32_dev(h, C.c_void_p(r.data_ptr()), m, 0, 0.0, C.byref(s)))
print("sum", repr(np.float32(s.value)), repr(np.add.reduce(rr)))
mean = np.float32(s.value) / np.float32(m)
print("mean", repr(mean), repr(np.mean(rr)))
_lib.check(lib.urhgpu_pairwise_sum_f32_dev(h, C.c_void_p(r.data_ptr()), m, 1, float(mean), C.byref(s)))
d = rr - np.mean(rr); d = d * d
print("sum2", repr(np.float32(s.value)), repr(np.add.reduce(d)))
print("var", repr(np.float32(s.value) / np.float32(m)), repr(np.var(rr)))
