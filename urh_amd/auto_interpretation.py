"""Drop-in mirror of the functions of the reference's `urh.cythonext.auto_interpretation` that sit on the IQ->bits path
(/root/reference/src/urh/cythonext/auto_interpretation.pyx), with the reference's exact signatures -- the names
`AutoInterpretation.py:8` and `Wavelet.py:3` import -- so that `setattr(urh.cythonext.auto_interpretation, name, ...)` (INTEGRATION.md
section 1) puts liburhgpu.so underneath the reference's own `AutoInterpretation.estimate`:

    segment_messages_from_magnitudes(magnitudes, noise_threshold)         :55-111
    get_threshold_divisor_histogram(plateau_lengths, threshold=0.2)        :113-143
    merge_plateaus(plateaus, tolerance, max_count)                         :145-176
    get_plateau_lengths(rect_data, center, percentage=25)                  :179-208
    median_filter(data, k=3)                                               :227-240

Host arrays in, host arrays out (as the Cython functions): the O(N) ones upload, run the HIP kernels and download; the two that work on
a few thousand plateau lengths are native host arithmetic inside the library, like the reference's.  The device-resident forms
(no PCIe traffic per call) are in urh_amd/estimators.py.  No CPU fallback: without liburhgpu.so or without a GPU the O(N) calls raise.
"""
import ctypes as C

import numpy as np

from . import _lib
from .signal_functions import _vp


def _floating_1d(a, name):
    """cython.floating[:]: a 1-D float32 or float64 buffer"""
    a = np.asarray(a)
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got {})".format(a.ndim))
    if a.dtype not in (np.float32, np.float64):
        raise TypeError("No matching signature found")               # what the fused-type dispatch raises
    return np.ascontiguousarray(a)


def segment_messages_from_magnitudes(magnitudes, noise_threshold: float, ctx=None) -> list:
    """List of (start, end) tuples of the stretches above the noise threshold, with the reference's 10-sample outlier tolerance."""
    m = _floating_1d(magnitudes, "magnitudes")
    n = len(m)
    if n == 0:
        return []
    ctx = ctx or _lib.default_context()
    cap = n // 20 + 3
    seg = np.empty((cap, 2), dtype=np.int64)
    n_seg = C.c_int64(0)
    _lib.check(_lib.load().urhgpu_segment_messages(ctx.handle, _vp(m), 1 if m.dtype == np.float64 else 0, n, float(noise_threshold),
                                                   _vp(seg), cap, C.byref(n_seg)))
    k = n_seg.value
    return list(zip(seg[:k, 0].tolist(), seg[:k, 1].tolist()))


def get_threshold_divisor_histogram(plateau_lengths, threshold: float = 0.2) -> np.ndarray:
    """uint64[max + 1]: histogram[v] = number of pairs (v, w >= v) of the given lengths whose ratio w / v has a fractional part below
    threshold.  (np.max of an empty array: ValueError, as in the reference.)"""
    p = np.ascontiguousarray(plateau_lengths, dtype=np.uint64)
    if p.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got {})".format(p.ndim))
    if len(p) == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    lib = _lib.load()
    hist_len = C.c_int64(0)
    _lib.check(lib.urhgpu_threshold_divisor_histogram(_vp(p), len(p), float(threshold), None, 0, C.byref(hist_len)))
    hist = np.zeros(hist_len.value, dtype=np.uint64)
    _lib.check(lib.urhgpu_threshold_divisor_histogram(_vp(p), len(p), float(threshold), _vp(hist), len(hist), C.byref(hist_len)))
    return hist


def merge_plateaus(plateaus, tolerance, max_count) -> np.ndarray:
    """Plateaus <= tolerance are glitches and are merged with their neighbours; at most max_count merged plateaus."""
    p = np.ascontiguousarray(plateaus, dtype=np.uint64)
    if len(p) == 0:
        return np.zeros(0, dtype=np.uint64)
    out = np.empty(len(p), dtype=np.uint64)
    n_out = C.c_int64(0)
    _lib.check(_lib.load().urhgpu_merge_plateaus(_vp(p), len(p), int(tolerance), int(max_count), _vp(out), C.byref(n_out)))
    return out[:n_out.value]


def get_plateau_lengths(rect_data, center, percentage: int = 25, ctx=None) -> np.ndarray:
    """Lengths (uint64) of the runs of (rect_data <= center) that start before `percentage` % of the signal."""
    x = np.asarray(rect_data)
    if x.ndim != 1 or x.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float'")
    if len(x) == 0 or center is None:
        return np.array([], dtype=np.uint64)
    x = np.ascontiguousarray(x)
    ctx = ctx or _lib.default_context()
    lib = _lib.load()
    cap = max(1 << 12, len(x) // 64)
    while True:
        out = np.empty(cap, dtype=np.uint64)
        n_out = C.c_int64(0)
        st = lib.urhgpu_get_plateau_lengths(ctx.handle, _vp(x), len(x), float(center), int(percentage), _vp(out), cap, C.byref(n_out))
        if st == _lib.ERR_CAPACITY:
            cap = n_out.value
            continue
        _lib.check(st)
        return out[:n_out.value].copy()


def median_filter(data, k: int = 3, ctx=None) -> np.ndarray:
    """float32[n]: element i is the upper median of data[i : i + k] (the window is cut at the end of the array)."""
    d = np.ascontiguousarray(data, dtype=np.float64)
    if d.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got {})".format(d.ndim))
    out = np.zeros(len(d), dtype=np.float32)
    if len(d) == 0:
        return out
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_median_filter(ctx.handle, _vp(d), len(d), int(k), _vp(out)))
    return out
