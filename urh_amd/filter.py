"""The spectrogram band-pass of the reference's `Filter` (row 6b of SURVEY.md §8a), backed by liburhgpu.so.

Mirrors /root/reference/src/urh/signalprocessing/Filter.py:
    get_filter_length_from_bandwidth :64-67, design_windowed_sinc_lpf :103-119, design_windowed_sinc_bandpass :121-131,
    apply_bandpass_filter :84-101 (np.convolve "same" or the FFT convolution :70-82 -- one centred linear convolution).
The taps are O(1/bw) host arithmetic in float64 / complex128 (the same numpy expressions, hence the same taps); the
O(N * taps) convolution runs on the GPU in fp64 (csrc/bandpass.hip).  numpy's summation order is not defined by the
reference, so the result is compared with a tolerance (tests/test_gpu_parity.py), not bit for bit.
No CPU fallback: without the library or a GPU every call raises.
"""
import ctypes as C
import math

import numpy as np

from . import _lib


def get_filter_length_from_bandwidth(bw) -> int:
    """Filter.py:64-67: ceil(4 / bw), forced odd"""
    n = int(math.ceil(4 / bw))
    return n + 1 if n % 2 == 0 else n


def get_bandwidth_from_filter_length(n):
    """Filter.py:60-62"""
    return 4 / n


def design_windowed_sinc_lpf(fc, bw) -> np.ndarray:
    """Filter.py:103-119: Blackman-windowed sinc normalised to unity gain (float64)"""
    n = get_filter_length_from_bandwidth(bw)
    h = np.sinc(2 * fc * (np.arange(n) - (n - 1) / 2.0))
    h = h * np.blackman(n)
    return h / np.sum(h)


def design_windowed_sinc_bandpass(f_low, f_high, bw) -> np.ndarray:
    """Filter.py:121-131: the low-pass shifted to the band centre (complex128)"""
    f_shift = (f_low + f_high) / 2
    f_c = (f_high - f_low) / 2
    n = get_filter_length_from_bandwidth(bw)
    return design_windowed_sinc_lpf(f_c, bw=bw) * np.exp(complex(0, 1) * np.pi * 2 * f_shift * np.arange(0, n, dtype=complex))


def bandpass_taps(f_low, f_high, filter_bw=0.08) -> np.ndarray:
    """The taps apply_bandpass_filter designs (Filter.py:86-92: swap, clip to +-0.5)"""
    if f_low > f_high:
        f_low, f_high = f_high, f_low
    f_low = max(-0.5, min(f_low, 0.5))
    f_high = max(-0.5, min(f_high, 0.5))
    return design_windowed_sinc_bandpass(f_low, f_high, filter_bw)


def _same_geometry(n: int, m: int):
    """(shift, n_out) of the reference's result for a capture of n samples and m taps (Filter.py:96-101)"""
    if n == 0:
        raise ValueError("math domain error")                 # math.log(math.sqrt(0)) in the reference (:96)
    if m < 8 * math.log(math.sqrt(n)):
        return (min(n, m) - 1) // 2, max(n, m)               # np.convolve(data, h, "same")
    # fft_convolve_1d: full[too_much : -too_much] with too_much = (m - 1) // 2
    too_much = (m - 1) // 2
    if too_much == 0:
        return 0, 0                                            # result[0:-0] is empty in the reference
    return too_much, n + m - 1 - 2 * too_much


def apply_bandpass_filter(data, f_low, f_high, filter_bw=0.08, ctx=None) -> np.ndarray:
    """Filter.apply_bandpass_filter (Filter.py:84-101) on host arrays: complex64[N] -> complex128[N]"""
    x = np.ascontiguousarray(np.asarray(data), dtype=np.complex64)
    h = np.ascontiguousarray(bandpass_taps(f_low, f_high, filter_bw), dtype=np.complex128)
    shift, n_out = _same_geometry(len(x), len(h))
    out = np.zeros(n_out, dtype=np.complex128)
    if n_out == 0:
        return out
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_bandpass(ctx.handle, C.c_void_p(x.ctypes.data), len(x), C.c_void_p(h.ctypes.data), len(h),
                                           shift, n_out, C.c_void_p(out.ctypes.data)))
    return out


def convolve_dev(pipe, iq, taps, shift, n_out, out_complex64=True, left=None, right=None):
    """out[i] = sum_k taps[k] * X(i + shift - k) on device memory (urhgpu_bandpass_dev).
    iq: complex64 (N,) or float32 (N, 2) tensor on pipe.device; taps: complex128 numpy array or device tensor;
    left / right: optional complex64 tensors that extend the capture (sharded captures) instead of zeros."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    if iq.dtype != torch.float32 or iq.dim() != 2 or iq.shape[1] != 2 or not iq.is_contiguous():
        raise ValueError("the band-pass takes a contiguous complex64 capture")
    if not torch.is_tensor(taps):
        taps = torch.from_numpy(np.ascontiguousarray(taps, dtype=np.complex128)).to(pipe.device)
    if taps.dtype != torch.complex128 or not taps.is_contiguous():
        raise ValueError("taps must be complex128")

    def edge(t):
        if t is None or t.numel() == 0:
            return None, 0
        if t.dtype == torch.complex64:
            t = torch.view_as_real(t)
        t = t.contiguous()
        return t, t.shape[0]

    left, n_left = edge(left)
    right, n_right = edge(right)
    out = torch.empty(n_out, dtype=torch.complex64 if out_complex64 else torch.complex128, device=pipe.device)
    pipe.ctx.set_stream(torch.cuda.current_stream(pipe.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_bandpass_dev(
        pipe.ctx.handle, C.c_void_p(iq.data_ptr()), iq.shape[0], C.c_void_p(taps.data_ptr()), taps.shape[0], shift, n_out,
        C.c_void_p(left.data_ptr()) if left is not None else None, n_left,
        C.c_void_p(right.data_ptr()) if right is not None else None, n_right,
        C.c_void_p(out.data_ptr()), 1 if out_complex64 else 0))
    pipe._bp_keep = (iq, taps, left, right)                   # alive until the next call (asynchronous launch)
    return out


def apply_bandpass_filter_dev(pipe, iq, f_low, f_high, filter_bw=0.08, out_complex64=True):
    """apply_bandpass_filter on a device-resident capture; out_complex64 fuses the cast SignalFrame applies to the result
    (/root/reference/src/urh/controller/widgets/SignalFrame.py:1578-1580)."""
    h = bandpass_taps(f_low, f_high, filter_bw)
    n = iq.shape[0]
    shift, n_out = _same_geometry(n, len(h))
    return convolve_dev(pipe, iq, h, shift, n_out, out_complex64)
