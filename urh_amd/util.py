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


def minmax(arr):
    """util.minmax (util.pyx:20-36): (min, max) of a small host array -- host arithmetic, as in the callers here."""
    if len(arr) == 0:
        return 0, 0
    a = np.asarray(arr)
    return a.min(), a.max()
