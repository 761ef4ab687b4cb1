"""Parameter estimators of the IQ->bits path (reference: src/urh/ainterpretation/AutoInterpretation.py) with the
O(N) passes on the GPU and the O(100)-element decisions on the host.

detect_noise_level  AutoInterpretation.py:60-91   (+ util.get_magnitudes, util.pyx:128-136)
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .signal_functions import dtype_code


def noise_chunks(n: int):
    """Chunk geometry of detect_noise_level (:65-72): chunks of max(1, int(n/100)) samples taken from the END of
    the capture backwards; the remainder at the front is dropped.  Returns (chunk, n_chunks)."""
    chunk = max(1, int(n * 1 / 100))
    return chunk, n // chunk


def noise_level_from_chunk_stats(sums, maxs, chunk: int) -> float:
    """The decision part of detect_noise_level on per-chunk (sum, max) of the magnitudes, chunk 0 = last chunk."""
    mean_values = (np.asarray(sums, dtype=np.float64) / chunk).astype(np.float32)      # np.mean -> float32 (:74-76)
    if len(mean_values) == 0:
        return 0
    minimum, maximum = mean_values.min(), mean_values.max()
    if maximum == 0 or minimum / maximum > 0.9:                                         # :77-80
        return 0
    idx = np.nonzero(mean_values <= 1.1 * np.min(mean_values))[0]                       # :83
    if len(idx) == 0:
        return 0
    result = np.max(np.asarray(maxs, dtype=np.float64)[idx])                            # :86
    return math.ceil(result * 10000) / 10000                                            # :91


def detect_noise_level_dev(pipe, iq) -> float:
    """detect_noise_level(get_magnitudes(iq)) for a capture resident on the GPU (`pipe`: DevicePipeline,
    `iq`: torch tensor (N, 2) or complex64 (N,)): one pass over the IQ stream, 2 x n_chunks doubles come back."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    from .pipeline import _torch_dtype
    n = int(iq.shape[0])
    if n <= 3:                                                                          # :61-62
        return 0
    chunk, n_chunks = noise_chunks(n)
    sums = torch.empty(n_chunks, dtype=torch.float64, device=iq.device)
    maxs = torch.empty(n_chunks, dtype=torch.float64, device=iq.device)
    pipe.ctx.set_stream(torch.cuda.current_stream(iq.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_magnitude_chunk_stats_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()),
                                                            dtype_code(_torch_dtype(iq)), n, chunk, n_chunks,
                                                            C.c_void_p(sums.data_ptr()), C.c_void_p(maxs.data_ptr())))
    return noise_level_from_chunk_stats(sums.cpu().numpy(), maxs.cpu().numpy(), chunk)
