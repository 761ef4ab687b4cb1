"""Parameter estimators of the IQ->bits path (reference: src/urh/ainterpretation/AutoInterpretation.py,
src/urh/cythonext/auto_interpretation.pyx) for a capture that lives in HBM.

Everything proportional to the capture, to the number of pulses or to the number of plateaus runs in liburhgpu.so: the passes over
the samples (magnitude statistics, segmentation, per-message center statistics, plateau boundaries, modulation features) as kernels
batched over all messages, the per-message integer decisions (tolerance, merged plateaus, divisor histogram, bit length) as native
host arithmetic on a thread pool.  What is left here is glue on a handful of numbers per message -- and numpy as the arbiter
wherever the reference's result is whatever numpy does with equal keys (np.argsort) or a borderline floating-point sum.

  detect_noise_level                         AutoInterpretation.py:60-91   (+ util.get_magnitudes, util.pyx:128-136)
  segment_messages / merge_..._for_ook       auto_interpretation.pyx:55-111, AutoInterpretation.py:107-148
  detect_center, get_plateau_lengths         AutoInterpretation.py:226-277, auto_interpretation.pyx:179-208
  tolerance / merge / round / bit length     AutoInterpretation.py:280-370, auto_interpretation.pyx:113-176
  detect_modulation(_for_messages)           AutoInterpretation.py:150-223
  estimate                                   AutoInterpretation.py:373-470
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


def fir_filter_detect_noise_dev(pipe, iq, taps, left=None):
    """Signal.filter_range over the whole capture followed by detect_noise_level (Signal.py:645-655, AutoInterpretation.py:60-91) in
    ONE pass over the samples: the magnitude chunk statistics of the filtered signal come out of the FIR kernel's epilogue
    (urhgpu_fir_filter_stats_dev).  iq: float32 (N, 2) or complex64 (N,) on the GPU; taps: complex64 (numpy or device).
    Returns (filtered capture, same shape / dtype as iq; noise threshold)."""
    torch = pipe.torch
    x = torch.view_as_real(iq) if iq.dtype == torch.complex64 else iq
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("FIR needs contiguous float32 / complex64 samples")
    if isinstance(taps, np.ndarray):
        taps = torch.from_numpy(np.ascontiguousarray(taps, dtype=np.complex64).view(np.float32).copy()).to(x.device)
    h = (torch.view_as_real(taps) if taps.dtype == torch.complex64 else taps).contiguous().reshape(-1, 2)
    n, m = int(x.shape[0]), int(h.shape[0])
    out = torch.empty_like(x)
    chunk, n_chunks = noise_chunks(n) if n > 3 else (0, 0)
    pipe.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
    lib, hdl = _lib.load(), pipe.ctx.handle
    lp = C.c_void_p(left.data_ptr()) if left is not None else None
    if n_chunks == 0 or m == 0:
        _lib.check(lib.urhgpu_fir_filter_dev(hdl, C.c_void_p(x.data_ptr()), n, C.c_void_p(h.data_ptr()), m, lp, C.c_void_p(out.data_ptr())))
        noise = detect_noise_level_dev(pipe, out)
    else:
        sums = torch.empty(n_chunks, dtype=torch.float64, device=x.device)
        maxs = torch.empty(n_chunks, dtype=torch.float64, device=x.device)
        _lib.check(lib.urhgpu_fir_filter_stats_dev(hdl, C.c_void_p(x.data_ptr()), n, C.c_void_p(h.data_ptr()), m, lp, C.c_void_p(out.data_ptr()),
                                                   chunk, n_chunks, C.c_void_p(sums.data_ptr()), C.c_void_p(maxs.data_ptr())))
        noise = noise_level_from_chunk_stats(sums.cpu().numpy(), maxs.cpu().numpy(), chunk)
    return (out if iq.dtype != torch.complex64 else torch.view_as_complex(out)), noise


# ======================================================================================================================
# Message segmentation, center, plateau lengths
# ======================================================================================================================
def _dev_f32(pipe, x):
    torch = pipe.torch
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 1 and x.is_contiguous()):
        raise ValueError("expected a contiguous float32 1-D tensor on the GPU")
    pipe.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
    return x


def segment_messages_dev(pipe, iq, noise_threshold: float, as_array: bool = False):
    """auto_interpretation.segment_messages_from_magnitudes (auto_interpretation.pyx:55-111) for a float32 capture on
    the GPU: list of (start, end).  The above/below-noise state machine with its 10-sample outlier tolerance is the
    run segmentation of the hot kernel with tolerance 9 on |sample| (urhgpu_segment_runs_dev); this entry point reads the rows back
    (one per state change) and applies the reference's index conventions on arrays -- estimate_dev uses message_ranges_dev, which
    keeps them on the device."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    n = int(iq.shape[0])
    if n == 0 or math.isnan(float(noise_threshold)):        # nothing compares greater than NaN: never above the noise
        return np.zeros((0, 2), np.int64) if as_array else []
    pipe.ctx.set_stream(torch.cuda.current_stream(iq.device).cuda_stream)
    cap = n // 10 + 2
    rows = torch.empty((cap, 2), dtype=torch.int64, device=iq.device)
    n_rows = torch.zeros(1, dtype=torch.int64, device=iq.device)
    from .pipeline import _torch_dtype
    _lib.check(_lib.load().urhgpu_segment_runs_dev(pipe.ctx.handle, C.c_void_p(iq.data_ptr()), dtype_code(_torch_dtype(iq)), n,
                                                   float(noise_threshold), C.c_void_p(rows.data_ptr()), cap,
                                                   C.c_void_p(n_rows.data_ptr())))
    r = rows[:int(n_rows.item())].cpu().numpy()
    tail = iq[max(0, n - 10):].cpu().numpy()
    if tail.dtype == np.float32:
        tail_mag = np.sqrt(tail[:, 0] * tail[:, 0] + tail[:, 1] * tail[:, 1]).astype(np.float64)   # fp32 sqrtf, as get_magnitudes
    else:                                                                              # C int arithmetic (wrapping), double sqrt
        a = tail.astype(np.int64)
        s32 = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        with np.errstate(invalid="ignore"):
            tail_mag = np.sqrt(s32.astype(np.float64))
    seg = _segments_array(r, n, tail_mag > float(np.float32(noise_threshold)))
    return seg if as_array else _as_tuples(seg)


def message_ranges_dev(pipe, iq, noise_threshold: float, merge: bool = True, cap_seg: int = 4096, cap_merged: int = 65536, qad_ask=None):
    """Segmentation and (optionally) the OOK merge without the per-pulse table leaving the GPU (urhgpu_message_ranges_dev).
    Returns (segments[:cap_seg], n_segments, merged[:cap_merged] or None, n_merged, ambiguous): (K, 2) int64 arrays as
    segment_messages_dev(as_array=True) / merge_message_segments_for_ook give them, truncated to the capacities.
    qad_ask: a float32 device tensor (n,) that the same pass fills with afp_demod(iq, noise_threshold, "ASK") (float32 captures:
    urhgpu_message_ranges_demod_dev)."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    n = int(iq.shape[0])
    pipe.ctx.set_stream(torch.cuda.current_stream(iq.device).cuda_stream)
    from .pipeline import _torch_dtype
    seg = np.empty((cap_seg, 2), np.int64)
    mrg = np.empty((cap_merged, 2), np.int64) if merge else None
    n_seg, n_mrg, amb = C.c_int64(0), C.c_int64(0), C.c_int(0)
    args = (pipe.ctx.handle, C.c_void_p(iq.data_ptr()), dtype_code(_torch_dtype(iq)), n, float(noise_threshold),
            C.c_void_p(seg.ctypes.data), cap_seg, C.byref(n_seg),
            C.c_void_p(mrg.ctypes.data) if merge else None, cap_merged if merge else 0, C.byref(n_mrg) if merge else None, C.byref(amb))
    if qad_ask is not None:
        if qad_ask.dtype != torch.float32 or qad_ask.numel() != n or not qad_ask.is_contiguous():
            raise ValueError("qad_ask: contiguous float32 tensor with one element per sample")
        _lib.check(_lib.load().urhgpu_message_ranges_demod_dev(*args, C.c_void_p(qad_ask.data_ptr())))
    else:
        _lib.check(_lib.load().urhgpu_message_ranges_dev(*args))
    return (seg[:min(n_seg.value, cap_seg)], n_seg.value, mrg[:min(n_mrg.value, cap_merged)] if merge else None, n_mrg.value,
            bool(amb.value))


def segments_from_rows(rows: np.ndarray, n: int, tail_above: np.ndarray):
    """rows: pulse table of the above(1)/below(0) states with tolerance 9 (row j: state BEFORE the j-th change, length);
    tail_above: above-noise flags of the last <= 10 samples.  Returns the reference's list of (start, end) tuples.

    Change k (k >= 1) happens at sample pos_k = len_0 - 1 + len_1 + ... + len_{k-1} (the run that triggered it started 10 samples
    earlier); a change to "above" opens a message at pos_k - 1 (auto_interpretation.pyx:101-104), a change to "below" closes the open
    one at pos_k - 1 (:95-99).  States alternate, so opens and closes pair up in order -- all of it array arithmetic."""
    return _as_tuples(_segments_array(rows, n, tail_above))


def _as_tuples(seg: np.ndarray):
    return list(zip(seg[:, 0].tolist(), seg[:, 1].tolist()))


def _segments_array(rows: np.ndarray, n: int, tail_above: np.ndarray) -> np.ndarray:
    """segments_from_rows as an int64 (K, 2) array (an OOK capture has one segment per pulse: hundreds of thousands)"""
    rows = np.asarray(rows, dtype=np.int64)
    states, lens = rows[:, 0], rows[:, 1]
    n_changes = len(rows) - 1
    pos = lens[0] - 1 + np.concatenate([[0], np.cumsum(lens[1:n_changes])]) if n_changes > 0 else np.zeros(0, np.int64)
    new_states = states[1:]
    opens = pos[new_states == 1] - 1
    closes = pos[new_states != 1] - 1
    if int(states[0]) == 1:                          # the capture starts above the noise: a message is open from sample 0
        opens = np.concatenate([[0], opens])
    seg = np.stack([opens[:len(closes)], closes], axis=1).astype(np.int64).reshape(-1, 2)
    if int(states[-1]) == 1:                         # still above at the end (:107-109): closes where the trailing below-run starts
        start = int(opens[-1]) if len(opens) else 0
        below = np.asarray(tail_above)[::-1]
        conseq_below = int(np.argmax(below)) if below.any() else len(below)
        if start < n - conseq_below:
            seg = np.concatenate([seg, np.array([[start, n - conseq_below]], dtype=np.int64)])
    return seg


def merge_message_segments_for_ook(segments: list):
    """AutoInterpretation.merge_message_segments_for_ook (AutoInterpretation.py:107-148): OOK pulses separated by pauses shorter
    than 8 x the (outlier-free) minimum pulse length belong to one message.  A merged message starts at its first pulse and is as
    long as its pulses and inner pauses together -- which telescopes to "ends where its last pulse ends"."""
    if len(segments) <= 1:
        return segments
    seg = np.asarray(segments, dtype=np.int64).reshape(-1, 2)
    pauses = (seg[1:, 0] - seg[:-1, 1]).astype(np.uint64)
    pulses = (seg[:, 1] - seg[:, 0]).astype(np.uint64)
    min_pulse_length = _inliers(pulses, 1).min()
    cut = np.nonzero(pauses >= 8 * min_pulse_length)[0] + 1           # a new message starts after every long pause
    first = np.concatenate([[0], cut])
    last = np.concatenate([cut, [len(seg)]]) - 1
    merged = np.stack([seg[first, 0], seg[last, 1]], axis=1)
    return merged if isinstance(segments, np.ndarray) else _as_tuples(merged)


def _inliers(data: np.ndarray, z: float) -> np.ndarray:
    """the values within z standard deviations of the mean (what max_ / min_without_outliers reduce over, AutoInterpretation.py:14-25)"""
    data = np.asarray(data)
    return data[np.abs(data - data.mean()) <= z * data.std()] if len(data) else data


def get_most_frequent_value(values: list):
    """AutoInterpretation.py:28-47: most frequent value, ties -> the LAST of the tied values in first-seen order."""
    if len(values) == 0:
        return None
    from collections import Counter
    ranked = Counter(values).most_common()
    top = ranked[0][1]
    return [v for v, c in ranked if c == top][-1]


def detect_center_dev(pipe, rect, max_size=None, _single=False):
    """AutoInterpretation.detect_center (AutoInterpretation.py:226-277) for a demodulated signal on the GPU.
    GPU passes: compaction rect > -4, min / max, np.var (float32 pairwise sums in numpy's order), histogram over the
    float64 edges np.arange(min, max + step, step), peak picking over the bins."""
    torch = pipe.torch
    x = _dev_f32(pipe, rect)
    n = int(x.shape[0])
    if max_size is None and n > 0 and not _single:
        # one "message": the batched pass (one read-back instead of five), unless its histogram does not fit the pool
        return centers_batched(pipe, x, [(0, n)])[0]
    lib, h = _lib.load(), pipe.ctx.handle
    kept = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=x.device)
    _lib.check(lib.urhgpu_compact_gt_dev(h, C.c_void_p(x.data_ptr()), n, -4.0, C.c_void_p(kept.data_ptr()), C.c_void_p(cnt.data_ptr())))
    k = int(cnt.item())
    a, b = int(0.05 * k), int(0.95 * k)                                                # :231
    r = kept[a:b]
    if max_size is not None and len(r) > max_size:                                    # :233-234
        r = r[0:max_size]
    m = int(r.shape[0])
    if m == 0:
        return None                 # np.var of an empty slice is nan -> np.arange raises ValueError -> None (:246-248)
    mm = torch.empty(2, dtype=torch.float32, device=x.device)
    _lib.check(lib.urhgpu_minmax_f32_dev(h, C.c_void_p(r.data_ptr()), m, C.c_void_p(mm.data_ptr())))
    hist_min, hist_max = (float(v) for v in mm.cpu().numpy())
    s = C.c_float(0.0)
    _lib.check(lib.urhgpu_pairwise_sum_f32_dev(h, C.c_void_p(r.data_ptr()), m, 0, 0.0, C.byref(s)))
    mean = np.float32(s.value) / np.float32(m)                                         # np.mean: float32 sum / float32 count
    _lib.check(lib.urhgpu_pairwise_sum_f32_dev(h, C.c_void_p(r.data_ptr()), m, 1, float(mean), C.byref(s)))
    hist_step = float(np.float32(s.value) / np.float32(m))                             # float(np.var(rect)) (:240)
    try:
        with np.errstate(all="ignore"):
            edges = np.arange(hist_min, hist_max + hist_step, hist_step)             # :243-245
        if len(edges) < 2:
            # np.histogram with fewer than 2 edges raises ValueError -> None (:246-248)
            return None
    except (ZeroDivisionError, ValueError):
        return None
    d_edges = torch.from_numpy(np.ascontiguousarray(edges, dtype=np.float64)).to(x.device)
    d_counts = torch.empty(len(edges) - 1, dtype=torch.int64, device=x.device)
    _lib.check(lib.urhgpu_histogram_f32_dev(h, C.c_void_p(r.data_ptr()), m, C.c_void_p(d_edges.data_ptr()), len(edges),
                                            C.c_void_p(d_counts.data_ptr())))
    return peaks_center(d_counts.cpu().numpy(), edges)


def _sliding_max(p: np.ndarray, w: int) -> np.ndarray:
    """m[j] = max(p[j : j + w]) for every full window, in O(len(p)) (block prefix / suffix maxima: van Herk, Gil-Werman)"""
    n = len(p)
    blocks = -(-n // w)
    q = np.full(blocks * w, np.iinfo(np.int64).min, dtype=np.int64)
    q[:n] = p
    b = q.reshape(blocks, w)
    pre = np.maximum.accumulate(b, axis=1).reshape(-1)
    suf = np.maximum.accumulate(b[:, ::-1], axis=1)[:, ::-1].reshape(-1)
    j = np.arange(n - w + 1)
    return np.maximum(suf[j], pre[j + w - 1])


def peaks_center(counts: np.ndarray, edges: np.ndarray):
    """detect_center's pick (AutoInterpretation.py:250-277) from a histogram: the two most populated bins that are strict maxima
    over +-(window - 1) bins (bins outside the histogram count as 0), mean of their left edges; None without such a bin.  Written
    on whole arrays: the peak mask from sliding-window maxima to either side (linear in the number of bins -- a nearly constant message
    has millions of them, and a comparison per bin AND offset took a minute there), candidates taken in the order np.argsort gives the
    bins (the reference's walk order, which is what decides between equally populated peaks)."""
    y = np.asarray(counts, dtype=np.int64)
    nb = len(y)
    if nb == 0:
        return None
    reach = max(2, int(0.05 * nb) + 1) - 1
    padded = np.concatenate([np.zeros(reach, np.int64), y, np.zeros(reach, np.int64)])
    side = _sliding_max(padded, reach)                    # side[j] = max(padded[j : j + reach]); bin i's left window starts at padded[i]
    peak = (y > 0) & (y > side[0:nb]) & (y > side[reach + 1:reach + 1 + nb])
    if not peak.any():
        return None
    walk = np.argsort(counts)[::-1]
    chosen = walk[peak[walk]][:2]
    return np.mean(np.asarray(edges)[chosen])


def get_plateau_lengths_dev(pipe, rect, center, percentage=25) -> np.ndarray:
    """auto_interpretation.get_plateau_lengths (auto_interpretation.pyx:179-208): lengths of the runs of
    (rect <= center) that START before `percentage` % of the signal and end inside it, uint64."""
    torch = pipe.torch
    x = _dev_f32(pipe, rect)
    n = int(x.shape[0])
    if n == 0 or center is None:
        return np.array([], dtype=np.uint64)
    limit = (percentage * n) // 100                               # C integer division (cdivision)
    # only the plateaus that START before `limit` count: all boundaries below it and the first one at or beyond it.  The
    # boundary pass runs over a window that is extended until it holds such a boundary (or the whole signal).
    w = min(n, limit + (1 << 16))
    while True:
        cap = min(w, max(1 << 16, w // 16))                       # plenty for real signals; retried with the exact count if not
        while True:
            idx = torch.empty(max(cap, 1), dtype=torch.int64, device=x.device)
            cnt = torch.zeros(1, dtype=torch.int64, device=x.device)
            _lib.check(_lib.load().urhgpu_edges_le_dev(pipe.ctx.handle, C.c_void_p(x.data_ptr()), w, float(center),
                                                       C.c_void_p(idx.data_ptr()), cap, C.c_void_p(cnt.data_ptr())))
            found = int(cnt.item())
            if found <= cap:
                break
            cap = found
        b = idx[:found].cpu().numpy()                             # run boundaries B_0 < B_1 < ...
        if w == n or (found and b[-1] >= limit):
            break
        w = min(n, 2 * w)
    # plateau k = [B_{k-1}, B_k) is appended at i = B_k if the sum appended so far (= B_{k-1}, or 0) is < limit
    starts = np.concatenate([[0], b[:-1]]) if len(b) else np.zeros(0, np.int64)
    keep = starts < limit
    lengths = (b - starts)[keep]
    return lengths.astype(np.uint64)


# ---- host decisions on plateau lengths (a few thousand integers per message) ---------------------------------------------
def merge_plateaus(plateaus, tolerance, max_count=10000) -> np.ndarray:
    """auto_interpretation.merge_plateaus (auto_interpretation.pyx:145-176): plateaus <= tolerance are glitches and are
    merged with their neighbours (looking ahead over alternating glitches); at most max_count merged plateaus.
    Sequential host arithmetic on a few thousand values: native code in the library (urhgpu_merge_plateaus), like the reference's."""
    p = np.ascontiguousarray(plateaus, dtype=np.uint64)
    out = np.empty(len(p), dtype=np.uint64)
    n_out = C.c_int64(0)
    _lib.check(_lib.load().urhgpu_merge_plateaus(C.c_void_p(p.ctypes.data), len(p), int(tolerance), int(max_count),
                                                 C.c_void_p(out.ctypes.data), C.byref(n_out)))
    return out[:n_out.value]


_MOD_LABELS = (None, "OOK", "ASK", "FSK", "PSK")


def detect_modulation_dev(pipe, iq, message_indices, wavelet_scale=4, median_filter_order=11, return_variances=False):
    """AutoInterpretation.detect_modulation (:150-205) for every given message of a capture on the GPU (float32 (N, 2) or complex64):
    compaction, Haar wavelet transforms through double-precision FFTs, median filter, variances and spectrum peaks on the device
    (urhgpu_detect_modulation_dev); returns the list of labels ("OOK" / "ASK" / "FSK" / "PSK" / None)."""
    torch = pipe.torch
    x = torch.view_as_real(iq) if iq.dtype == torch.complex64 else iq
    if x.dtype != torch.float32:
        from .iq_array import convert_to
        x = convert_to(x, np.float32, pipe.ctx)                   # IQArray.as_complex64 (IQArray.py:92-93)
    x = x.contiguous()
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    n_msgs = len(ranges)
    labels = np.zeros(max(n_msgs, 1), dtype=np.int32)
    variances = np.zeros((max(n_msgs, 1), 4), dtype=np.float64)
    pipe.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_detect_modulation_dev(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]),
                                                        C.c_void_p(ranges.ctypes.data), n_msgs, int(wavelet_scale), int(median_filter_order),
                                                        C.c_void_p(labels.ctypes.data), C.c_void_p(variances.ctypes.data)))
    out = [_MOD_LABELS[int(v)] for v in labels[:n_msgs]]
    return (out, variances[:n_msgs]) if return_variances else out


def detect_modulation_for_messages_dev(iq, message_indices: list, pipe=None):
    """AutoInterpretation.detect_modulation_for_messages (:208-223): the most common label of the first 100 messages, classified on
    the GPU (urhgpu_detect_modulation_dev); among equally common labels the one that occurs first wins, as max() over the list does
    in the reference (:50-57)."""
    if pipe is None:
        from .pipeline import DevicePipeline
        pipe = DevicePipeline(iq.device.index)
    labels = [m for m in detect_modulation_dev(pipe, iq, list(message_indices[0:100])) if m is not None]
    if not labels:
        return None
    votes = {}
    for position, label in enumerate(labels):
        count, first = votes.get(label, (0, position))
        votes[label] = (count + 1, first)
    return min(votes, key=lambda label: (-votes[label][0], votes[label][1]))


def centers_batched(pipe, data, message_indices, max_bins: int = 4096):
    """detect_center (AutoInterpretation.py:226-277) of every message in one batched device pass (urhgpu_msg_center_stats):
    statistics, histograms and the peak picking; one read-back of a few numbers per message.  A message whose histogram has more
    than max_bins bins (a nearly constant signal: tiny variance) goes through the single-message path; one whose result depends on
    np.argsort's order of equal counts is decided by numpy on its histogram.  Returns a list with a float or None per message."""
    arr = centers_array(pipe, data, message_indices, max_bins)
    return [None if c != c else np.float64(c) for c in arr.tolist()]


def centers_array(pipe, data, message_indices, max_bins: int = 4096) -> np.ndarray:
    """centers_batched as a float64 array, NaN where a message has no center (a center itself is never NaN: it is a bin edge)"""
    x = _dev_f32(pipe, data)
    n_msgs = len(message_indices)
    if n_msgs == 0:
        return np.zeros(0, np.float64)
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    stats = np.zeros((n_msgs, 8), dtype=np.float64)
    cen = np.zeros(n_msgs, dtype=np.float64)
    flag = np.zeros(n_msgs, dtype=np.int32)
    lib = _lib.load()
    _lib.check(lib.urhgpu_msg_center_stats(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), C.c_void_p(ranges.ctypes.data),
                                           n_msgs, max_bins, C.c_void_p(stats.ctypes.data), None, C.c_void_p(cen.ctypes.data),
                                           C.c_void_p(flag.ctypes.data)))
    return _settle_centers(pipe, x, ranges, cen, flag, max_bins)


def centers_and_decisions(pipe, data, message_indices, percentage: int = 25, max_bins: int = 4096):
    """(centers float64 with NaN = none, tolerance, bit_length) of every message from ONE native call (urhgpu_msg_estimate: center
    statistics, then plateau boundaries and length counts with the centers still on the device); tolerance / bit_length as
    _plateau_decisions gives them.  Messages whose center the host has to settle (more bins than the pool holds, a tie numpy decides)
    get their second stage in a call of their own; a capture whose histogram pool does not fit one batch takes the two calls."""
    x = _dev_f32(pipe, data)
    n_msgs = len(message_indices)
    if n_msgs == 0:
        return np.zeros(0, np.float64), np.zeros(0, np.int64), np.zeros(0, np.int64)
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    stats = np.zeros((n_msgs, 8), dtype=np.float64)
    cen = np.zeros(n_msgs, dtype=np.float64)
    flag = np.zeros(n_msgs, dtype=np.int32)
    tol = np.zeros(n_msgs, dtype=np.int64)
    bl = np.zeros(n_msgs, dtype=np.int64)
    st = _lib.load().urhgpu_msg_estimate(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), C.c_void_p(ranges.ctypes.data), n_msgs, max_bins,
                                         int(percentage), 1 << 16, C.c_void_p(stats.ctypes.data), C.c_void_p(cen.ctypes.data),
                                         C.c_void_p(flag.ctypes.data), C.c_void_p(tol.ctypes.data), C.c_void_p(bl.ctypes.data))
    if st == _lib.ERR_UNSUPPORTED:
        centers = centers_array(pipe, x, ranges, max_bins)
        t2, b2 = _plateau_decisions(pipe, x, ranges, centers.astype(np.float32).astype(np.float64), percentage)
        return centers, t2, b2
    _lib.check(st)
    late = np.nonzero((flag == 2) | (flag == 3))[0]
    centers = _settle_centers(pipe, x, ranges, cen, flag, max_bins)
    if len(late):
        sub = np.ascontiguousarray(ranges[late])
        t2, b2 = _plateau_decisions(pipe, x, sub, centers[late].astype(np.float32).astype(np.float64), percentage)
        tol[late], bl[late] = t2, b2
    return centers, tol, bl


def _settle_centers(pipe, x, ranges, cen, flag, max_bins):
    """the centers of a batch as a float64 array (NaN: none), the messages the device could not settle (flag 2, 3) decided here"""
    lib = _lib.load()
    centers = np.where(flag == 1, cen, np.nan)

    def put(m, c):
        centers[m] = np.nan if c is None else c
    for m in np.nonzero(flag == 2)[0].tolist():      # more bins than the pool holds (a nearly constant message): the single-message path
        put(m, detect_center_dev(pipe, x[int(ranges[m, 0]):int(ranges[m, 1])], _single=True))
    ties = np.nonzero(flag == 3)[0]
    # equally populated peaks: np.argsort's order of equal keys decides.  Only THOSE messages' histograms are fetched, a bounded
    # number at a time (a capture can hold 10^5 .. 10^6 messages; max_bins int64 counters each on the host)
    for t0 in range(0, len(ties), _TIE_BATCH):
        tie = ties[t0:t0 + _TIE_BATCH]
        tr = np.ascontiguousarray(ranges[tie])
        tstats = np.zeros((len(tie), 8), dtype=np.float64)
        hist = np.zeros((len(tie), max_bins), dtype=np.int64)
        _lib.check(lib.urhgpu_msg_center_stats(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), C.c_void_p(tr.ctypes.data),
                                               len(tie), max_bins, C.c_void_p(tstats.ctypes.data), C.c_void_p(hist.ctypes.data), None, None))
        for k, m in enumerate(tie.tolist()):
            n_edges = int(tstats[k, 6])
            hist_min, hist_max, step = float(tstats[k, 2]), float(tstats[k, 3]), float(tstats[k, 5])
            with np.errstate(all="ignore"):
                edges = np.arange(hist_min, hist_max + step, step)                  # the same edges the device binned with
            if len(edges) != n_edges or edges[0] != tstats[k, 7]:                   # cannot happen; never bin against other edges silently
                put(m, detect_center_dev(pipe, x[int(ranges[m, 0]):int(ranges[m, 1])], _single=True))
            else:
                put(m, peaks_center(hist[k, :n_edges - 1], edges))
    return centers


_TIE_BATCH = 1024


def _plateaus_raw(pipe, x, ranges, cen, percentage):
    """urhgpu_msg_plateaus: (lens, off) -- message m's plateau lengths are lens[|off[m]| : |off[m + 1]|), a negative end offset
    -(end + 1) marks a message whose first search window held no boundary beyond the percentage mark."""
    n_msgs = len(ranges)
    off = np.zeros(n_msgs + 1, dtype=np.int64)
    cap = int(max(1 << 16, (ranges[:, 1] - ranges[:, 0]).sum() // 64))
    lib = _lib.load()
    while True:
        lens = np.empty(cap, dtype=np.uint64)
        st = lib.urhgpu_msg_plateaus(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), C.c_void_p(ranges.ctypes.data),
                                     C.c_void_p(cen.ctypes.data), n_msgs, int(percentage), 1 << 16, C.c_void_p(off.ctypes.data),
                                     C.c_void_p(lens.ctypes.data), cap)
        if st == _lib.ERR_CAPACITY:
            cap = int(off[n_msgs])
            continue
        _lib.check(st)
        return lens, off


def _plateau_decisions(pipe, x, ranges, cen, percentage):
    """urhgpu_msg_plateau_decisions: (tolerance, bit_length) int64 arrays over the messages (-1 None, -2 numpy's order decides, -3 the
    search window was too small), the plateau lengths never leaving the GPU unless a message's glitches need their order"""
    n_msgs = len(ranges)
    tol = np.zeros(n_msgs, dtype=np.int64)
    bl = np.zeros(n_msgs, dtype=np.int64)
    _lib.check(_lib.load().urhgpu_msg_plateau_decisions(
        pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), C.c_void_p(ranges.ctypes.data), C.c_void_p(cen.ctypes.data),
        n_msgs, int(percentage), 1 << 16, C.c_void_p(tol.ctypes.data), C.c_void_p(bl.ctypes.data)))
    return tol, bl


def plateau_lengths_batched(pipe, data, message_indices, centers, percentage: int = 25):
    """get_plateau_lengths (auto_interpretation.pyx:179-208) of every message that has a center: one batched device pass
    (urhgpu_msg_plateaus), one read-back.  Returns a list of uint64 arrays (empty for messages without a center)."""
    x = _dev_f32(pipe, data)
    n_msgs = len(message_indices)
    if n_msgs == 0:
        return []
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    cen = np.array([np.nan if c is None else float(np.float32(c)) for c in centers], dtype=np.float64)
    lens, off = _plateaus_raw(pipe, x, ranges, cen, percentage)
    out = []
    begin = 0
    for m in range(n_msgs):
        end = int(off[m + 1])
        if end < 0:                                # no boundary beyond the 25 % mark inside the first window: single-message path
            end = -end - 1
            out.append(get_plateau_lengths_dev(pipe, x[int(ranges[m, 0]):int(ranges[m, 1])], centers[m], percentage))
        else:
            out.append(lens[begin:end].copy())
        begin = end
    return out


def _bit_length_with_numpy_order(plateau_lengths):
    """(tolerance or None, bit_length or None) of a message whose divisor histogram holds equal counts: the library builds the
    histogram (urhgpu_msg_divisor_histogram), numpy orders it -- the reference's result is whatever np.argsort does with equal keys --
    and the library applies the selection rule to that order (urhgpu_bit_length_from_order)."""
    lib = _lib.load()
    p = np.ascontiguousarray(plateau_lengths, dtype=np.uint64)
    hist_len, tol = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.urhgpu_msg_divisor_histogram(C.c_void_p(p.ctypes.data), len(p), None, 0, C.byref(hist_len), C.byref(tol)))
    tolerance = None if tol.value < 0 else tol.value
    if hist_len.value < 0:
        return tolerance, None
    hist = np.zeros(hist_len.value, dtype=np.uint64)
    _lib.check(lib.urhgpu_msg_divisor_histogram(C.c_void_p(p.ctypes.data), len(p), C.c_void_p(hist.ctypes.data), len(hist),
                                                C.byref(hist_len), C.byref(tol)))
    order = np.ascontiguousarray(np.argsort(hist)[::-1], dtype=np.int64)
    out = C.c_int64(0)
    _lib.check(lib.urhgpu_bit_length_from_order(C.c_void_p(hist.ctypes.data), C.c_void_p(order.ctypes.data), len(hist), C.byref(out)))
    return tolerance, out.value


def bit_length_of_message(plateau_lengths):
    """(tolerance or None, bit_length or None) of one message from its plateau lengths: glitch tolerance, merged plateaus,
    divisor histogram (AutoInterpretation.py:416-433)."""
    return bit_lengths_batched([np.asarray(plateau_lengths, dtype=np.uint64)])[0]


def _bit_lengths_raw(lens, off, plateaus_of):
    """urhgpu_msg_bit_lengths on the (lens, off) layout of _plateaus_raw; plateaus_of(m): message m's plateau lengths (for the
    messages numpy has to decide)."""
    n_msgs = len(off) - 1
    tol = np.zeros(n_msgs, dtype=np.int64)
    bl = np.zeros(n_msgs, dtype=np.int64)
    lens = lens if len(lens) else np.zeros(1, np.uint64)
    _lib.check(_lib.load().urhgpu_msg_bit_lengths(C.c_void_p(lens.ctypes.data), C.c_void_p(off.ctypes.data), n_msgs,
                                                  C.c_void_p(tol.ctypes.data), C.c_void_p(bl.ctypes.data)))
    out = [(None if t < 0 else t, None if b < 0 else b) for t, b in zip(tol.tolist(), bl.tolist())]
    for m in np.nonzero((bl == -2) | (tol == -2))[0].tolist():
        out[m] = _bit_length_with_numpy_order(np.array(plateaus_of(m), dtype=np.uint64))
    return out


def bit_lengths_batched(all_plateaus):
    """[(tolerance or None, bit_length or None)] for every message from its plateau lengths: the native batch call
    (urhgpu_msg_bit_lengths: tolerance, merged plateaus, rounded lengths, divisor histogram from the value multiset, decision); a
    message whose decision hangs on how np.argsort orders equal counts gets its histogram ordered by numpy (_bit_length_with_numpy_order)."""
    n_msgs = len(all_plateaus)
    if n_msgs == 0:
        return []
    off = np.zeros(n_msgs + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(p) for p in all_plateaus])
    lens = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint64) for p in all_plateaus]) if off[-1] else np.zeros(1, np.uint64))
    return _bit_lengths_raw(lens, off, lambda m: all_plateaus[m])


def estimate_dev(pipe, iq, noise: float = None, modulation: str = None, timings: dict = None, keep: dict = None):
    """AutoInterpretation.estimate (AutoInterpretation.py:373-470) for a float32 capture resident on the GPU: noise threshold,
    message ranges (segmentation + OOK merge on the device), modulation vote over the first 100 messages (device), demodulation, then
    three batched calls over ALL messages -- center statistics with peak picking, plateau boundaries, bit-length decisions -- and the
    vote over the messages.  Per capture the host sees a few numbers per message.  keep (a dict): receives the demodulated signal
    (device tensor) and the parameters it was demodulated with, so that the caller can slice it instead of demodulating again."""
    from .pipeline import DemodParams
    import time
    torch = pipe.torch
    t_last = [time.perf_counter()]

    def lap(name):                                   # stage wall times for bench.py's breakdown (timings given: synchronises)
        if timings is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            timings[name] = round((now - t_last[0]) * 1e3, 3)
            t_last[0] = now
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    noise = detect_noise_level_dev(pipe, iq) if noise is None else noise
    lap("noise_ms")
    # one row per OOK pulse before merging: hundreds of thousands, which stay on the GPU -- the host gets the first segments
    # (modulation detection looks at 100) and the merged messages
    # an OOK / ASK capture named as such is demodulated by the segmentation pass itself (the same samples, the same threshold): one pass
    # over the capture instead of two
    data = None
    if modulation in ("OOK", "ASK") and iq.dtype == torch.float32 and int(iq.shape[0]) > 0 and float(noise) == float(noise):
        data = torch.empty(int(iq.shape[0]), dtype=torch.float32, device=iq.device)
    segments, n_segments, merged, n_merged, ambiguous = message_ranges_dev(pipe, iq, noise, qad_ask=data)
    lap("segment_messages_ms")
    if modulation is None:
        modulation = detect_modulation_for_messages_dev(iq, segments[:100].tolist(), pipe=pipe)
        if modulation is None:
            return None
    if modulation == "OOK":
        if ambiguous:                                # a pulse length within rounding of mean +- std: decide in numpy's summation order
            message_indices = merge_message_segments_for_ook(segment_messages_dev(pipe, iq, noise, as_array=True))
        elif n_merged > len(merged):
            message_indices = message_ranges_dev(pipe, iq, noise, cap_seg=1, cap_merged=n_merged)[2]
        else:
            message_indices = merged
    elif n_segments > len(segments):
        message_indices = message_ranges_dev(pipe, iq, noise, merge=False, cap_seg=n_segments)[0]
    else:
        message_indices = segments
    if modulation in ("OOK", "ASK"):
        mod = "ASK"
    elif modulation in ("FSK", "PSK"):
        mod = modulation
    else:
        raise ValueError("Unsupported Modulation")
    lap("modulation_and_merge_ms")
    if data is None:
        data = pipe.afp_demod(iq, DemodParams(mod, 1, float(noise)))
    if keep is not None:                             # the demodulated signal and what it was demodulated with: the caller's Signal.qad cache
        keep.update(qad=data, mod=mod, noise=float(noise))
    lap("afp_demod_ms")
    # centers (float64, NaN = no center) and the plateau decisions of every message: ONE native call (round 5; two until then, with the
    # centers crossing PCIe twice in between)
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    x32 = _dev_f32(pipe, data)
    center_of, tol_raw, bl_raw = centers_and_decisions(pipe, x32, ranges, 25)
    has_center = center_of == center_of
    lap("centers_and_plateaus_ms")
    if (tol_raw == -3).any():                        # a message whose first plateau outlasts the search window: the per-message path
        all_centers = [None if c != c else np.float64(c) for c in center_of.tolist()]
        decisions = bit_lengths_batched(plateau_lengths_batched(pipe, data, message_indices, all_centers))
        tol_of = np.array([-1 if t is None else t for t, _ in decisions], dtype=np.int64)
        len_of = np.array([-1 if b is None else b for _, b in decisions], dtype=np.int64)
    else:
        tol_of, len_of = tol_raw.copy(), bl_raw.copy()
        for m in np.nonzero((bl_raw == -2) | (tol_raw == -2))[0].tolist():      # numpy's order of equal histogram counts decides
            t, b = _bit_length_with_numpy_order(
                get_plateau_lengths_dev(pipe, x32[int(ranges[m, 0]):int(ranges[m, 1])], None if not has_center[m] else np.float64(center_of[m]), 25))
            tol_of[m] = -1 if t is None else t
            len_of[m] = -1 if b is None else b
        tol_of[tol_of < 0] = -1
        len_of[len_of < 0] = -1
    # the votes (AutoInterpretation.py:407-470) on arrays: a message with a center contributes its tolerance; it votes for a center and
    # a bit length when its merged plateaus gave a bit length above tolerance + 1
    tolerances = tol_of[has_center & (tol_of >= 0)]
    votes = has_center & (len_of >= 0) & (len_of > np.maximum(tol_of, 0) + 1)
    lap("bit_lengths_host_ms")
    if not votes.any():
        return None
    if modulation in ("OOK", "ASK"):
        center = _inliers(center_of[votes], 2).min()         # min_without_outliers(centers, z=2)
    else:
        center = np.mean(center_of[votes])
    bit_length = get_most_frequent_value(len_of[votes].tolist())
    if len(tolerances):
        # np.percentile(tolerances, 50) (:462), written out (numpy's own routine costs 60 us on a hundred values): linear interpolation
        # between the two middle order statistics, in numpy's form b - (b - a) * (1 - t) for t >= 0.5; exact for these integers
        v = np.sort(tolerances).astype(np.float64)
        idx = 0.5 * (len(v) - 1)
        lo, hi = int(np.floor(idx)), int(np.ceil(idx))
        g = idx - lo
        tolerance = v[lo] + (v[hi] - v[lo]) * g if g < 0.5 else v[hi] - (v[hi] - v[lo]) * (1.0 - g)
    else:                                                # no message had a tolerance (np.percentile of an empty list: IndexError in the reference's numpy)
        tolerance = max(1, int(0.05 * bit_length))
    return {"modulation_type": "ASK" if modulation == "OOK" else modulation, "bit_length": bit_length, "center": center,
            "tolerance": int(tolerance), "noise": noise}
