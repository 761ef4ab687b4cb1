"""Drop-in mirror of the functions of the reference's `urh.cythonext.util` that sit on the IQ->bits path
(/root/reference/src/urh/cythonext/util.pyx), backed by liburhgpu.so.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from .signal_functions import _DT, _iq, _vp


def get_magnitudes(arr, ctx=None) -> np.ndarray:
    """util.get_magnitudes (util.pyx:128-136): float64[N] = sqrt(I*I + Q*Q) in the element type's arithmetic."""
    a = _iq(arr)
    out = np.zeros(len(a), dtype=np.float64)
    if len(a) == 0:
        return out
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_get_magnitudes(ctx.handle, _vp(a), _DT[a.dtype], len(a), _vp(out)))
    return out


def minmax(arr, ctx=None):
    """util.minmax (util.pyx:20-36): (min, max) of a 1-D array of one of the fused `iq` element types (int8 / uint8 / int16 / uint16 /
    float32), as Python scalars; (0, 0) for an empty array.  detect_center calls it on a whole message (AutoInterpretation.py:236), so
    it is a pass over the samples: one reduction kernel with the reference's comparisons (urhgpu_minmax)."""
    a = np.asarray(arr)
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got {})".format(a.ndim))
    if a.dtype not in _DT:
        raise TypeError("No matching signature found")                  # what the fused-type dispatch raises
    if len(a) == 0:
        return 0, 0
    a = np.ascontiguousarray(a)
    out = np.zeros(2, dtype=a.dtype)
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_minmax(ctx.handle, _vp(a), _DT[a.dtype], len(a), _vp(out)))
    return out[0].item(), out[1].item()
