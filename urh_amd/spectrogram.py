"""Mirror of the reference's `urh.signalprocessing.Spectrogram.Spectrogram`
(/root/reference/src/urh/signalprocessing/Spectrogram.py) with the numeric part on the GPU (spectrogram.hip):

    stft(samples)                      :94-116   complex128 (frames, window_size)
    calculate_spectrogram(samples)     :158-164  float32 decibels (frames, window_size): fftshift, complex64, arr2decibel, fliplr
    apply_bgra_lookup / image arrays   :196-210  BGRA bytes (window_size, frames, 4) -- what the reference wraps in a QImage

`samples` is a numpy complex64 array or a torch tensor in HBM (complex64, or float32 (N, 2)); with a device tensor
`calculate_spectrogram(..., device=True)` returns a device tensor and nothing crosses PCIe.  Same properties as the
reference class (window_size, overlap_factor, window_function, hop_size, time_bins, freq_bins, data_min / data_max).
"""
import ctypes as C
import math

import numpy as np

from . import _lib


class Spectrogram(object):
    MAX_LINES_PER_VIEW = 1000
    DEFAULT_FFT_WINDOW_SIZE = 1024

    def __init__(self, samples, window_size=DEFAULT_FFT_WINDOW_SIZE, overlap_factor=0.5, window_function=np.hanning, ctx=None):
        self.samples = samples
        self.window_size = window_size
        self.overlap_factor = overlap_factor
        self.window_function = window_function
        self.data_min, self.data_max = -140, 10
        self._ctx = ctx

    # ---- properties of the reference class ------------------------------------------------------------------
    @property
    def hop_size(self):
        return self.window_size - int(self.overlap_factor * self.window_size)          # :87-92

    @property
    def time_bins(self):
        return int(math.ceil(len(self.samples) / self.hop_size))                       # :77-79

    @property
    def freq_bins(self):
        return self.window_size

    # ---- device plumbing ------------------------------------------------------------------------------------
    def _device_samples(self, samples):
        import torch
        if isinstance(samples, np.ndarray):
            a = samples
            if a.dtype != np.complex64:
                if a.ndim == 2 and a.shape[1] == 2:                                    # IQArray.as_complex64 of float data
                    a = np.ascontiguousarray(a, dtype=np.float32).view(np.complex64).reshape(-1)
                else:
                    a = a.astype(np.complex64)
            t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        else:
            t = samples
        if t.dtype == torch.complex64:
            t = torch.view_as_real(t)
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 2:
            raise ValueError("samples must be complex64 (N,) or float32 (N, 2)")
        return t.contiguous()

    def _launch(self, samples, want_stft):
        import torch
        ws, hop = int(self.window_size), int(self.hop_size)
        x = self._device_samples(samples)
        n = x.shape[0]
        frames = max(1, (max(n, ws) - ws) // hop + 1)                                   # :101-104 (zero padding up to one window)
        ctx = self._ctx or _lib.default_context()
        window = torch.from_numpy(np.ascontiguousarray(self.window_function(ws), dtype=np.float64)).to(x.device)
        tw = torch.from_numpy(np.exp(-2j * np.pi * np.arange(ws // 2) / ws).astype(np.complex128).view(np.float64)).to(x.device)
        stft = torch.empty((frames, ws), dtype=torch.complex128, device=x.device) if want_stft else None
        db = torch.empty((frames, ws), dtype=torch.float32, device=x.device) if not want_stft else None
        ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_spectrogram_dev(ctx.handle, C.c_void_p(x.data_ptr()), n, ws, hop, frames,
                                                      C.c_void_p(window.data_ptr()), C.c_void_p(tw.data_ptr()),
                                                      C.c_void_p(stft.data_ptr()) if want_stft else None,
                                                      C.c_void_p(db.data_ptr()) if not want_stft else None))
        return stft if want_stft else db

    # ---- the reference's methods ----------------------------------------------------------------------------
    def stft(self, samples=None):
        """Short-time Fourier transform, complex128 (frames, window_size), as Spectrogram.stft (:94-116)."""
        return self._launch(self.samples if samples is None else samples, True).cpu().numpy()

    def calculate_spectrogram(self, samples=None, device=False):
        """Spectrogram.__calculate_spectrogram (:158-164): decibels, float32 (frames, window_size)."""
        db = self._launch(self.samples if samples is None else samples, False)
        return db if device else db.cpu().numpy()

    def apply_bgra_lookup(self, data, colormap, data_min=None, data_max=None, device=False):
        """Spectrogram.apply_bgra_lookup (:196-210) with normalize=True; colormap: (n, 4) uint8 BGRA.  Returns
        (window_size, frames, 4) uint8."""
        import torch
        if data_min is None or data_max is None:
            raise ValueError("Can't normalize without data min and data max")
        d = data if not isinstance(data, np.ndarray) else torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).cuda()
        frames, ws = d.shape
        cm = torch.from_numpy(np.ascontiguousarray(colormap, dtype=np.uint8).view(np.uint32).reshape(-1).astype(np.int32)).to(d.device)
        img = torch.empty((ws, frames), dtype=torch.int32, device=d.device)
        ctx = self._ctx or _lib.default_context()
        ctx.set_stream(torch.cuda.current_stream(d.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_bgra_lookup_dev(ctx.handle, C.c_void_p(d.data_ptr()), frames, ws, C.c_void_p(cm.data_ptr()),
                                                      cm.numel(), float(data_min), float(data_max), C.c_void_p(img.data_ptr())))
        if device:
            return img
        return img.cpu().numpy().view(np.uint8).reshape(ws, frames, 4)

    def create_spectrogram_image_array(self, colormap, sample_start=None, sample_end=None, step=None):
        """Spectrogram.create_spectrogram_image (:166-183) up to the QImage: BGRA bytes (window_size, frames, 4)."""
        s = self.samples[sample_start:sample_end:step]
        if not isinstance(s, np.ndarray):
            s = s.contiguous()
        return self.apply_bgra_lookup(self.calculate_spectrogram(s, device=True), colormap, self.data_min, self.data_max)
