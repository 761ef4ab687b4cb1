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
# Message segmentation, center, plateau lengths: O(N) passes on the GPU, decisions on the host
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
    run segmentation of the hot kernel with tolerance 9 on |sample| (urhgpu_segment_runs_dev); the host walks the
    resulting rows (one per state change) and applies the reference's index conventions."""
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


def message_ranges_dev(pipe, iq, noise_threshold: float, merge: bool = True, cap_seg: int = 4096, cap_merged: int = 65536):
    """Segmentation and (optionally) the OOK merge without the per-pulse table leaving the GPU (urhgpu_message_ranges_dev).
    Returns (segments[:cap_seg], n_segments, merged[:cap_merged] or None, n_merged, ambiguous): (K, 2) int64 arrays as
    segment_messages_dev(as_array=True) / merge_message_segments_for_ook give them, truncated to the capacities."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    n = int(iq.shape[0])
    pipe.ctx.set_stream(torch.cuda.current_stream(iq.device).cuda_stream)
    from .pipeline import _torch_dtype
    seg = np.empty((cap_seg, 2), np.int64)
    mrg = np.empty((cap_merged, 2), np.int64) if merge else None
    n_seg, n_mrg, amb = C.c_int64(0), C.c_int64(0), C.c_int(0)
    _lib.check(_lib.load().urhgpu_message_ranges_dev(
        pipe.ctx.handle, C.c_void_p(iq.data_ptr()), dtype_code(_torch_dtype(iq)), n, float(noise_threshold),
        seg.ctypes.data_as(C.c_void_p), cap_seg, C.byref(n_seg),
        mrg.ctypes.data_as(C.c_void_p) if merge else None, cap_merged if merge else 0, C.byref(n_mrg) if merge else None, C.byref(amb)))
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
    min_pulse_length = min_without_outliers(pulses, z=1)
    cut = np.nonzero(pauses >= 8 * min_pulse_length)[0] + 1           # a new message starts after every long pause
    first = np.concatenate([[0], cut])
    last = np.concatenate([cut, [len(seg)]]) - 1
    merged = np.stack([seg[first, 0], seg[last, 1]], axis=1)
    return merged if isinstance(segments, np.ndarray) else _as_tuples(merged)


def max_without_outliers(data: np.ndarray, z=3):
    """AutoInterpretation.py:14-18"""
    if len(data) == 0:
        return None
    return np.max(data[abs(data - np.mean(data)) <= z * np.std(data)])


def min_without_outliers(data: np.ndarray, z=2):
    """AutoInterpretation.py:21-25"""
    if len(data) == 0:
        return None
    return np.min(data[abs(data - np.mean(data)) <= z * np.std(data)])


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
    return center_from_histogram(d_counts.cpu().numpy(), edges)


def center_from_histogram(y: np.ndarray, x: np.ndarray):
    """The peak picking of detect_center (AutoInterpretation.py:250-277): up to two bins, most populated first, that are
    strict maxima over +-(window-1) bins; the center is the mean of their left edges."""
    num_values = 2
    window_size = max(2, int(0.05 * len(y)) + 1)
    levels = []
    ny = len(y)
    for index in np.argsort(y)[::-1]:
        lo, hi = max(0, index - (window_size - 1)), min(ny, index + window_size)
        around = np.concatenate([y[lo:index], y[index + 1:hi]])
        # neighbours outside the histogram count as 0, so a bin with count 0 is never a strict maximum
        if y[index] > 0 and (len(around) == 0 or y[index] > around.max()):
            levels.append(x[index])
        if len(levels) == num_values:
            break
    if len(levels) == 0:
        return None
    return np.mean(levels)


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
def estimate_tolerance_from_plateau_lengths(plateau_lengths, relative_max=0.05):
    """AutoInterpretation.py:280-298: the largest "tiny" plateau length, i.e. below 5 % of the outlier-free maximum."""
    if len(plateau_lengths) <= 1:
        return None
    unique = np.unique(plateau_lengths)
    limit = relative_max * max_without_outliers(unique, z=2)
    if unique[0] > 1 and unique[0] >= limit:
        return 0
    result = 0
    for value in unique:
        if value > 1 and value >= limit:
            break
        result = value
    return result


def merge_plateaus(plateaus, tolerance, max_count=10000) -> np.ndarray:
    """auto_interpretation.merge_plateaus (auto_interpretation.pyx:145-176): plateaus <= tolerance are glitches and are
    merged with their neighbours (looking ahead over alternating glitches); at most max_count merged plateaus.
    Sequential host arithmetic on a few thousand values: native code in the library (urhgpu_merge_plateaus), like the reference's."""
    p = np.ascontiguousarray(plateaus, dtype=np.uint64)
    out = np.empty(len(p), dtype=np.uint64)
    n_out = C.c_int64(0)
    _lib.check(_lib.load().urhgpu_merge_plateaus(p.ctypes.data_as(C.c_void_p), len(p), int(tolerance), int(max_count),
                                                 out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[:n_out.value]


def merge_plateau_lengths(plateau_lengths, tolerance=None):
    """AutoInterpretation.py:301-310"""
    if tolerance is None:
        tolerance = estimate_tolerance_from_plateau_lengths(plateau_lengths)
    if tolerance == 0 or tolerance is None:
        return plateau_lengths
    return merge_plateaus(plateau_lengths, tolerance, max_count=10000)


def round_plateau_lengths(plateau_lengths):
    """AutoInterpretation.py:313-326 (in place): round to the leading digits, e.g. 99 -> 100, 293 -> 300.  The number of kept
    digits is the median decimal length (at most 3); int(round(p / f)) * f with Python's round = half-to-even on the double quotient."""
    p = np.asarray(plateau_lengths, dtype=np.uint64)
    digit_counts = np.searchsorted(_POW10, p, side="right") + 1                 # len(str(p))
    n_digits = min(3, int(np.percentile(digit_counts, 50)))
    f = 10 ** (n_digits - 1)
    plateau_lengths[:] = (np.rint(p / f).astype(np.uint64) * np.uint64(f)).astype(np.asarray(plateau_lengths).dtype)


_POW10 = np.array([10 ** k for k in range(1, 20)], dtype=np.uint64)


def get_threshold_divisor_histogram(plateau_lengths, threshold=0.2) -> np.ndarray:
    """auto_interpretation.get_threshold_divisor_histogram (auto_interpretation.pyx:113-143): histogram[min(x, y)] += 1
    for every pair (i < j) whose ratio max / min has a fractional part below `threshold` (float32 threshold, double ratio).

    The reference walks all P^2 / 2 pairs; the outcome of a pair depends on its two VALUES only, and after
    round_plateau_lengths there are few distinct ones, so the histogram is assembled from the value multiset:
    c_a * c_b pairs for distinct values a < b that pass the test, c_a * (c_a - 1) / 2 pairs of equal values (ratio 1)."""
    p = np.asarray(plateau_lengths, dtype=np.uint64)
    hist = np.zeros(int(np.max(p)) + 1, dtype=np.uint64)
    vals, counts = np.unique(p[p != 0], return_counts=True)
    if len(vals) == 0:
        return hist
    thr = float(np.float32(threshold))
    c = counts.astype(np.uint64)
    hist[vals.astype(np.int64)] += c * (c - np.uint64(1)) // np.uint64(2)
    if len(vals) > 1:
        lo = vals[:, None]                                     # vals ascending: row a < column b above the diagonal
        hi = vals[None, :]
        frac = hi.astype(np.float64) / lo.astype(np.float64) - (hi // lo).astype(np.float64)
        ok = np.triu(frac < thr, k=1)
        hist[vals.astype(np.int64)] += (ok * c[None, :]).sum(axis=1, dtype=np.uint64) * c
    return hist


def get_bit_length_from_plateau_lengths(merged_plateau_lengths) -> int:
    """AutoInterpretation.py:344-370"""
    if len(merged_plateau_lengths) == 0:
        return 0
    if len(merged_plateau_lengths) == 1:
        return int(merged_plateau_lengths[0])
    round_plateau_lengths(merged_plateau_lengths)
    histogram = get_threshold_divisor_histogram(merged_plateau_lengths)
    if len(histogram) == 0:
        return 0
    sorted_indices = np.argsort(histogram)[::-1]
    max_count = histogram[sorted_indices[0]]
    result = sorted_indices[0]
    for i in range(1, len(sorted_indices)):
        if histogram[sorted_indices[i]] < 0.25 * max_count:
            break
        if sorted_indices[i] <= 0.5 * result:
            result = sorted_indices[i]
    return int(result)


# ---- modulation detection (host, like the reference: numpy on the first 100 messages) -----------------------------
def median_filter(data, k: int = 3) -> np.ndarray:
    """auto_interpretation.median_filter (auto_interpretation.pyx:213-240): float32 result; the window of sample i is
    data[i : i + k] cut at the end of the array (`start` is computed and ignored, :233-238), values rounded to float32
    before the sort, result = sorted[k' // 2]."""
    x = np.asarray(data, dtype=np.float64).astype(np.float32)
    n = len(x)
    out = np.zeros(n, dtype=np.float32)
    if n == 0:
        return out
    k = int(k)
    full = n - k + 1
    if full > 0:
        win = np.lib.stride_tricks.sliding_window_view(x, k)
        out[:full] = np.sort(win, axis=1)[:, k // 2]
    for i in range(max(full, 0), n):
        w = np.sort(x[i:n])
        out[i] = w[len(w) // 2]
    return out


def normalized_haar_wavelet(omega, scale):
    """Wavelet.normalized_haar_wavelet (Wavelet.py:7-12)"""
    omega_cpy = omega[:] / scale
    omega_cpy[0] = 1.0
    return (1j * np.square(-1 + np.exp(0.5j * omega))) / omega_cpy


def cwt_haar(x: np.ndarray, scale=10):
    """Wavelet.cwt_haar (Wavelet.py:15-43)"""
    next_power_two = 2 ** int(np.log2(len(x)))
    x = x[0:next_power_two]
    num_data = len(x)
    x_hat = np.fft.fft(x)
    f = 2.0 * np.pi / num_data
    omega = f * np.concatenate((np.arange(0, num_data // 2), np.arange(num_data // 2, num_data) * -1))
    psi_hat = np.sqrt(2.0 * np.pi * scale) * normalized_haar_wavelet(scale * omega, scale)
    W = np.fft.ifft(x_hat * psi_hat)
    return W[2 * scale:-2 * scale]


def detect_modulation(data: np.ndarray, wavelet_scale=4, median_filter_order=11):
    """AutoInterpretation.detect_modulation (AutoInterpretation.py:150-205) for ONE message (complex64 samples on the host)."""
    n_data = len(data)
    data = data[np.abs(data) > 0]
    if len(data) == 0:
        return None
    if n_data - len(data) > 3:
        return "OOK"
    data = data / np.abs(np.max(data))
    mag_wavlt = np.abs(cwt_haar(data, scale=wavelet_scale))
    if len(mag_wavlt) == 0:
        return None
    norm_mag_wavlt = np.abs(cwt_haar(data / np.abs(data), scale=wavelet_scale))
    var_mag = np.var(mag_wavlt)
    var_norm_mag = np.var(norm_mag_wavlt)
    var_filtered_mag = np.var(median_filter(mag_wavlt, k=median_filter_order))
    var_filtered_norm_mag = np.var(median_filter(norm_mag_wavlt, k=median_filter_order))
    if all(v < 0.15 for v in (var_mag, var_norm_mag, var_filtered_mag, var_filtered_norm_mag)):
        return "OOK"
    if var_mag > 1.5 * var_norm_mag:
        return "ASK"
    if var_mag > 10 * var_filtered_mag:
        return "PSK"
    fft = np.fft.fft(data[0:2 ** int(np.log2(len(data)))])
    fft = np.abs(np.fft.fftshift(fft))
    ten_greatest_indices = np.argsort(fft)[::-1][0:10]
    greatest_index = ten_greatest_indices[0]
    min_distance = 10
    min_freq = 100
    if any(abs(i - greatest_index) >= min_distance and fft[i] >= min_freq for i in ten_greatest_indices):
        return "FSK"
    return "OOK"


def most_common(values: list):
    """AutoInterpretation.most_common (:50-57): ties go to the value that appears first"""
    from collections import Counter
    counter = Counter(values)
    return max(values, key=counter.get)


def _as_complex64(iq_host: np.ndarray) -> np.ndarray:
    """IQArray.as_complex64 (IQArray.py:92-93) = convert_to(np.float32) (:127-185) viewed as complex64: integer captures are
    scaled with the reference's float32 operations (multiply by 1/128 or 1/32768, unsigned types then add -1)."""
    a = iq_host
    if a.dtype == np.float32:
        f = a
    elif a.dtype == np.uint8:
        f = np.add(np.multiply(a, 1 / 128, dtype=np.float32), -1.0, dtype=np.float32)
    elif a.dtype == np.int8:
        f = np.multiply(a, 1 / 128, dtype=np.float32)
    elif a.dtype == np.uint16:
        f = np.add(np.multiply(a, 1 / 32768, dtype=np.float32), -1.0, dtype=np.float32)
    elif a.dtype == np.int16:
        f = np.multiply(a, 1 / 32768, dtype=np.float32)
    else:
        raise ValueError("Unsupported dtype")
    return np.ascontiguousarray(f).flatten(order="C").view(np.complex64)


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
                                                        ranges.ctypes.data_as(C.c_void_p), n_msgs, int(wavelet_scale), int(median_filter_order),
                                                        labels.ctypes.data_as(C.c_void_p), variances.ctypes.data_as(C.c_void_p)))
    out = [_MOD_LABELS[int(v)] for v in labels[:n_msgs]]
    return (out, variances[:n_msgs]) if return_variances else out


def detect_modulation_for_messages_dev(iq, message_indices: list, pipe=None):
    """AutoInterpretation.detect_modulation_for_messages (:208-223): the most common label of the first 100 messages.  With a
    pipeline the messages are classified on the GPU; without one they are copied to the host and classified with numpy."""
    max_messages = 100
    if pipe is not None:
        mods = [m for m in detect_modulation_dev(pipe, iq, list(message_indices[0:max_messages])) if m is not None]
    else:
        mods = []
        for start, end in message_indices[0:max_messages]:
            mod = detect_modulation(_as_complex64(iq[start:end].cpu().numpy()))
            if mod is not None:
                mods.append(mod)
    if len(mods) == 0:
        return None
    return most_common(mods)


def centers_batched(pipe, data, message_indices, max_bins: int = 4096):
    """detect_center (AutoInterpretation.py:226-277) of every message in one batched device pass (urhgpu_msg_center_stats):
    statistics, histograms and the peak picking; one read-back of a few numbers per message.  A message whose histogram has more
    than max_bins bins (a nearly constant signal: tiny variance) goes through the single-message path; one whose result depends on
    np.argsort's order of equal counts is decided by numpy on its histogram.  Returns a list with a float or None per message."""
    x = _dev_f32(pipe, data)
    n_msgs = len(message_indices)
    if n_msgs == 0:
        return []
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    stats = np.zeros((n_msgs, 8), dtype=np.float64)
    cen = np.zeros(n_msgs, dtype=np.float64)
    flag = np.zeros(n_msgs, dtype=np.int32)
    lib = _lib.load()
    _lib.check(lib.urhgpu_msg_center_stats(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), ranges.ctypes.data_as(C.c_void_p),
                                           n_msgs, max_bins, stats.ctypes.data_as(C.c_void_p), None, cen.ctypes.data_as(C.c_void_p),
                                           flag.ctypes.data_as(C.c_void_p)))
    centers = [np.float64(c) if f == 1 else None for c, f in zip(cen.tolist(), flag.tolist())]
    hist = None
    for m in np.nonzero(flag >= 2)[0].tolist():
        if flag[m] == 2:
            centers[m] = detect_center_dev(pipe, x[int(ranges[m, 0]):int(ranges[m, 1])], _single=True)
            continue
        if hist is None:                             # equally populated peaks: fetch the histograms, np.argsort decides
            hist = np.zeros((n_msgs, max_bins), dtype=np.int64)
            _lib.check(lib.urhgpu_msg_center_stats(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), ranges.ctypes.data_as(C.c_void_p),
                                                   n_msgs, max_bins, stats.ctypes.data_as(C.c_void_p), hist.ctypes.data_as(C.c_void_p), None, None))
        n_edges = int(stats[m, 6])
        hist_min, hist_max, step = float(stats[m, 2]), float(stats[m, 3]), float(stats[m, 5])
        with np.errstate(all="ignore"):
            edges = np.arange(hist_min, hist_max + step, step)                  # the same edges the device binned with
        if len(edges) != n_edges or edges[0] != stats[m, 7]:                    # cannot happen; never bin against other edges silently
            centers[m] = detect_center_dev(pipe, x[int(ranges[m, 0]):int(ranges[m, 1])], _single=True)
        else:
            centers[m] = center_from_histogram(hist[m, :n_edges - 1], edges)
    return centers


def _plateaus_raw(pipe, x, ranges, cen, percentage):
    """urhgpu_msg_plateaus: (lens, off) -- message m's plateau lengths are lens[|off[m]| : |off[m + 1]|), a negative end offset
    -(end + 1) marks a message whose first search window held no boundary beyond the percentage mark."""
    n_msgs = len(ranges)
    off = np.zeros(n_msgs + 1, dtype=np.int64)
    cap = int(max(1 << 16, (ranges[:, 1] - ranges[:, 0]).sum() // 64))
    lib = _lib.load()
    while True:
        lens = np.empty(cap, dtype=np.uint64)
        st = lib.urhgpu_msg_plateaus(pipe.ctx.handle, C.c_void_p(x.data_ptr()), int(x.shape[0]), ranges.ctypes.data_as(C.c_void_p),
                                     cen.ctypes.data_as(C.c_void_p), n_msgs, int(percentage), 1 << 16, off.ctypes.data_as(C.c_void_p),
                                     lens.ctypes.data_as(C.c_void_p), cap)
        if st == _lib.ERR_CAPACITY:
            cap = int(off[n_msgs])
            continue
        _lib.check(st)
        return lens, off


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


def bit_length_of_message(plateau_lengths):
    """(tolerance or None, bit_length or None) of one message from its plateau lengths: glitch tolerance, merged plateaus,
    divisor histogram (AutoInterpretation.py:416-433)."""
    tolerance = estimate_tolerance_from_plateau_lengths(plateau_lengths)
    merged = merge_plateau_lengths(plateau_lengths, tolerance=0 if tolerance is None else tolerance)
    if len(merged) < 2:
        return tolerance, None
    return tolerance, get_bit_length_from_plateau_lengths(merged)


def _bit_lengths_raw(lens, off, plateaus_of):
    """urhgpu_msg_bit_lengths on the (lens, off) layout of _plateaus_raw; plateaus_of(m): message m's plateau lengths (for the
    messages numpy has to decide)."""
    n_msgs = len(off) - 1
    tol = np.zeros(n_msgs, dtype=np.int64)
    bl = np.zeros(n_msgs, dtype=np.int64)
    lens = lens if len(lens) else np.zeros(1, np.uint64)
    _lib.check(_lib.load().urhgpu_msg_bit_lengths(lens.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), n_msgs,
                                                  tol.ctypes.data_as(C.c_void_p), bl.ctypes.data_as(C.c_void_p)))
    out = [(None if t < 0 else t, None if b < 0 else b) for t, b in zip(tol.tolist(), bl.tolist())]
    for m in np.nonzero((bl == -2) | (tol == -2))[0].tolist():
        out[m] = bit_length_of_message(np.array(plateaus_of(m), dtype=np.uint64))
    return out


def bit_lengths_batched(all_plateaus):
    """[(tolerance or None, bit_length or None)] for every message from its plateau lengths: the native batch call
    (urhgpu_msg_bit_lengths: tolerance, merged plateaus, rounded lengths, divisor histogram from the value multiset, decision); a
    message whose decision hangs on how np.argsort orders equal counts is repeated in numpy (bit_length_of_message)."""
    n_msgs = len(all_plateaus)
    if n_msgs == 0:
        return []
    off = np.zeros(n_msgs + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(p) for p in all_plateaus])
    lens = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint64) for p in all_plateaus]) if off[-1] else np.zeros(1, np.uint64))
    return _bit_lengths_raw(lens, off, lambda m: all_plateaus[m])


def estimate_dev(pipe, iq, noise: float = None, modulation: str = None, timings: dict = None):
    """AutoInterpretation.estimate (AutoInterpretation.py:373-470) for a float32 capture resident on the GPU.  Device passes over
    the samples: magnitude statistics, segmentation, demodulation, then TWO batched passes over all messages (center statistics +
    histograms; plateau boundaries for the chosen centers).  The host sees a histogram of a few dozen bins and a few thousand
    plateau lengths per message and takes the decisions the reference takes (peak picking, tolerance, merged plateaus, divisor
    histogram from the multiset of rounded lengths), plus detect_modulation on the first 100 messages like the reference."""
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
    segments, n_segments, merged, n_merged, ambiguous = message_ranges_dev(pipe, iq, noise)
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
    data = pipe.afp_demod(iq, DemodParams(mod, 1, float(noise)))
    lap("afp_demod_ms")
    all_centers = centers_batched(pipe, data, message_indices)
    lap("centers_ms")
    ranges = np.ascontiguousarray(message_indices, dtype=np.int64).reshape(-1, 2)
    cen = np.array([np.nan if c is None else float(np.float32(c)) for c in all_centers], dtype=np.float64)
    lens, off = _plateaus_raw(pipe, _dev_f32(pipe, data), ranges, cen, 25) if len(ranges) else (np.zeros(1, np.uint64), np.zeros(1, np.int64))
    if (off < 0).any():                              # a message whose first plateau outlasts the search window: the per-message path
        all_plateaus = plateau_lengths_batched(pipe, data, message_indices, all_centers)
        lap("plateaus_ms")
        decisions = bit_lengths_batched(all_plateaus)
    else:
        lap("plateaus_ms")
        decisions = _bit_lengths_raw(lens, off, lambda m: lens[int(off[m]):int(off[m + 1])])
    centers, bit_lengths, tolerances = [], [], []
    for center, (tolerance, bit_length) in zip(all_centers, decisions):
        if center is None:
            continue
        if tolerance is not None:
            tolerances.append(tolerance)
        if bit_length is not None and bit_length > (tolerance or 0) + 1:
            centers.append(center)
            bit_lengths.append(bit_length)
    lap("bit_lengths_host_ms")
    if modulation in ("OOK", "ASK"):
        center = min_without_outliers(np.array(centers), z=2)
        if center is None:
            return None
    elif len(centers) > 0:
        center = np.mean(centers)
    else:
        return None
    bit_length = get_most_frequent_value(bit_lengths)
    if bit_length is None:
        return None
    try:
        tolerance = np.percentile(tolerances, 50)
    except IndexError:
        tolerance = max(1, int(0.05 * bit_length))
    return {"modulation_type": "ASK" if modulation == "OOK" else modulation, "bit_length": bit_length, "center": center,
            "tolerance": int(tolerance), "noise": noise}
