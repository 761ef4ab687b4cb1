"""Mirror of the numeric part of the reference's `urh.cythonext.path_creator`
(/root/reference/src/urh/cythonext/path_creator.pyx): the per-pixel minimum / maximum pass over a 1-D signal
(`create_path`, :19-82) runs on the GPU (plot.hip, urhgpu_path_minmax); what it hands to Qt stays on the host.

    create_path_arrays(samples, start, end, subpath_ranges)  -> [(x int64, values)]   the arguments of array_to_QPath (:80)
    path_bytes(x, values)                                    -> bytes                 the QDataStream form array_to_QPath builds
                                                                                       (:101-129): `QDataStream(QByteArray(b)) >> QPainterPath()`
`samples` is a numpy array (staged through the C ABI) or a torch tensor that already lives in HBM (e.g. the demodulated
signal of a DevicePipeline pass): only 2 values per pixel come back.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

PIXELS_PER_PATH = 5000          # urh.settings.PIXELS_PER_PATH (/root/reference/src/urh/settings.py:35)
_DT = {np.dtype(np.int8): _lib.DT_I8, np.dtype(np.uint8): _lib.DT_U8, np.dtype(np.int16): _lib.DT_I16,
       np.dtype(np.uint16): _lib.DT_U16, np.dtype(np.float32): _lib.DT_F32}


def _minmax(samples, start, end, spp, ctx):
    pixels = (end - start + spp - 1) // spp
    lib = _lib.load()
    if isinstance(samples, np.ndarray):
        a = np.ascontiguousarray(samples)
        if a.dtype not in _DT:
            raise ValueError("Unsupported dtype")
        values = np.zeros(2 * pixels, dtype=a.dtype)
        _lib.check(lib.urhgpu_path_minmax(ctx.handle, C.c_void_p(a.ctypes.data), _DT[a.dtype], len(a), start, end, spp,
                                          C.c_void_p(values.ctypes.data)))
        return values
    import torch                                     # device-resident signal
    from .pipeline import _torch_dtype
    dt = _torch_dtype(samples)
    values = torch.empty(2 * pixels, dtype=samples.dtype, device=samples.device)
    ctx.set_stream(torch.cuda.current_stream(samples.device).cuda_stream)
    _lib.check(lib.urhgpu_path_minmax_dev(ctx.handle, C.c_void_p(samples.data_ptr()), _DT[dt], start, end, spp,
                                          C.c_void_p(values.data_ptr())))
    return values.cpu().numpy()


def create_path_arrays(samples, start: int, end: int, subpath_ranges=None, ctx=None, pixels_on_path: int = PIXELS_PER_PATH):
    start, end = int(start), int(end)
    num_samples = end - start
    subpath_ranges = [(start, end)] if subpath_ranges is None else subpath_ranges
    spp = (abs(num_samples) // pixels_on_path) * (1 if num_samples >= 0 else -1)     # C division of two long long (:38)
    if spp > 1:
        rng = np.arange(start, end, spp, dtype=np.int64)
        scale_factor = float(np.float32(num_samples / (2.0 * len(rng))))              # `cdef float` (:24, :48)
        values = _minmax(samples, start, end, spp, ctx or _lib.default_context())
        x = np.repeat(rng, 2)
    else:
        x = np.arange(start, end, dtype=np.int64)
        values = samples[start:end]
        if not isinstance(values, np.ndarray):
            values = values.cpu().numpy()
        scale_factor = 1.0
    if scale_factor == 0:
        scale_factor = 1                                                              # :73-74
    out = []
    for r in subpath_ranges:                                                          # :76-81
        s0 = ((((r[0] - start) / scale_factor) * scale_factor) - 2 * scale_factor) / scale_factor
        s0 = int(max(0, math.floor(s0)))
        s1 = ((((r[1] - start) / scale_factor) * scale_factor) + 2 * scale_factor) / scale_factor
        s1 = int(max(0, math.ceil(s1)))
        out.append((x[s0:s1], values[s0:s1]))
    return out


def path_bytes(x, values) -> bytes:
    """array_to_QPath's buffer (:101-129): numVerts(i4) 0(i4)? no -- numVerts(i4), then per vertex c(i4)=1, x(f8), y(f8) with
    y = -values (np.negative: unsigned types wrap, as in the reference), then cStart(i4)=0, fillRule(i4)=0; big endian."""
    n = len(x)
    if n == 0:
        return b""
    buf = bytearray(4 + n * 20 + 8)
    buf[0:4] = int(n).to_bytes(4, "big", signed=True)
    arr = np.frombuffer(buf, dtype=[("c", ">i4"), ("x", ">f8"), ("y", ">f8")], count=n, offset=4)
    arr["x"] = x
    arr["y"] = np.negative(values)
    arr["c"] = 1
    return bytes(buf)
