"""numpy restatements of the reference's per-message decision functions (AutoInterpretation.py, auto_interpretation.pyx,
Wavelet.py) -- TEST INFRASTRUCTURE: what the gpu tests compare the device / native implementations with where the reference itself
is not available (the GPU box), and what tests/test_estimators_host.py pins against the real reference where it is.  The product
(urh_amd/estimators.py) does not import this file."""
import ctypes as C

import numpy as np

from urh_amd import _lib


def merge_plateaus(plateaus, tolerance, max_count=10000) -> np.ndarray:
    p = np.ascontiguousarray(plateaus, dtype=np.uint64)
    out = np.empty(len(p), dtype=np.uint64)
    n_out = C.c_int64(0)
    _lib.check(_lib.load().urhgpu_merge_plateaus(p.ctypes.data_as(C.c_void_p), len(p), int(tolerance), int(max_count),
                                                 out.ctypes.data_as(C.c_void_p), C.byref(n_out)))
    return out[:n_out.value]


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
