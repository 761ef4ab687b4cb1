"""Python face of the CPU oracle (oracle/urh_oracle.c + numpy restatements).

TEST INFRASTRUCTURE ONLY -- see the header of urh_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Function names/signatures mirror the reference's Cython modules so that a parity test reads
`oracle.afp_demod(...) == urh_amd.signal_functions.afp_demod(...)`.
"""
import array
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "liburh_oracle.so")
_SRC = os.path.join(HERE, "urh_oracle.c")

DT_CODES = {np.dtype(np.int8): 0, np.dtype(np.uint8): 1, np.dtype(np.int16): 2,
            np.dtype(np.uint16): 3, np.dtype(np.float32): 4}
MOD_CODES = {"ASK": 0, "FSK": 1, "PSK": 2}


def build(force: bool = False) -> str:
    """gcc -O2 -ffp-contract=off: the reference's x86-64 build has no FMA contraction."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=gnu11",
                               _SRC, "-o", _SO, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_grab_pulse_lens.restype = C.c_int64
        _lib.orc_ppseq_to_bits.restype = C.c_int64
        _lib.orc_afp_demod.restype = C.c_int
        _lib.orc_modulate.restype = C.c_int64
    return _lib


def noise_for_mod_type(mod_type: str) -> float:
    """signal_functions.pyx:31-44"""
    if mod_type == "ASK":
        return 0.0
    if mod_type in ("FSK", "PSK", "OQPSK"):
        return -4.0
    if mod_type == "QAM":
        return -0.0
    return 0.0


def _mod_args(mod_type: str):
    if mod_type in MOD_CODES:
        return MOD_CODES[mod_type], 0.0
    return 3, noise_for_mod_type(mod_type)


def _iq2d(samples):
    a = np.ascontiguousarray(samples)
    if a.ndim != 2 or a.shape[1] != 2 or a.dtype not in DT_CODES:
        raise ValueError("Unsupported dtype")
    return a


def atan2f(y, x) -> np.ndarray:
    """elementwise host-libm atan2f"""
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(y)
    lib().orc_atan2f_array(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int64(len(y)),
                           out.ctypes.data_as(C.c_void_p))
    return out


def get_magnitudes(arr) -> np.ndarray:
    a = _iq2d(arr)
    out = np.zeros(len(a), dtype=np.float64)
    lib().orc_get_magnitudes(a.ctypes.data_as(C.c_void_p), DT_CODES[a.dtype], C.c_int64(len(a)),
                             out.ctypes.data_as(C.c_void_p))
    return out


def afp_demod(samples, noise_mag: float, mod_type: str, mod_order: int, costas_loop_bandwidth: float = 0.1):
    a = _iq2d(samples)
    n = len(a)
    out = np.zeros(n, dtype=np.float32)
    mod, sentinel = _mod_args(mod_type)
    rc = lib().orc_afp_demod(a.ctypes.data_as(C.c_void_p), DT_CODES[a.dtype], C.c_int64(n), C.c_float(noise_mag),
                             mod, int(mod_order), C.c_float(costas_loop_bandwidth), C.c_float(sentinel),
                             out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("Unsupported dtype")
    return out


def get_center_thresholds(center: float, spacing: float, modulation_order: int) -> np.ndarray:
    out = np.empty(max(modulation_order - 1, 0), dtype=np.float32)
    lib().orc_get_center_thresholds(C.c_float(center), C.c_float(spacing), int(modulation_order),
                                    out.ctypes.data_as(C.c_void_p))
    return out


def grab_pulse_lens(samples, center: float, tolerance: int, modulation_type: str, samples_per_symbol: int,
                    bits_per_symbol: int = 1, center_spacing: float = 0.1) -> np.ndarray:
    s = np.ascontiguousarray(samples, dtype=np.float32)
    n = len(s)
    rows = np.zeros((max(n, 1), 2), dtype=np.int64)
    mod, sentinel = _mod_args(modulation_type)
    k = lib().orc_grab_pulse_lens(s.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_float(center),
                                  C.c_uint16(tolerance), mod, C.c_uint32(samples_per_symbol),
                                  C.c_uint8(bits_per_symbol), C.c_float(center_spacing), C.c_float(sentinel),
                                  rows.ctypes.data_as(C.c_void_p))
    return rows[:k].copy()


def fir_filter(input_samples, filter_taps) -> np.ndarray:
    x = np.ascontiguousarray(input_samples, dtype=np.complex64)
    h = np.ascontiguousarray(filter_taps, dtype=np.complex64)
    out = np.zeros(len(x), dtype=np.complex64)
    lib().orc_fir_filter(x.ctypes.data_as(C.c_void_p), C.c_int64(len(x)), h.ctypes.data_as(C.c_void_p),
                         C.c_int64(len(h)), out.ctypes.data_as(C.c_void_p))
    return out


def iir_filter(a, b, signal) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.ascontiguousarray(signal, dtype=np.complex64)
    out = np.zeros(len(x), dtype=np.complex64)
    lib().orc_iir_filter(a.ctypes.data_as(C.c_void_p), C.c_int64(len(a)), b.ctypes.data_as(C.c_void_p),
                         C.c_int64(len(b)), x.ctypes.data_as(C.c_void_p), C.c_int64(len(x)),
                         out.ctypes.data_as(C.c_void_p))
    return out


def ppseq_to_bits_flat(ppseq, samples_per_symbol: int, bits_per_symbol: int, write_bit_sample_pos=True,
                       pause_threshold=8):
    """ProtocolAnalyzer.py:323-414 with flat outputs: (bits u8, msg_off i64, pauses i64, pos i64, pos_off i64)."""
    pp = np.ascontiguousarray(ppseq, dtype=np.int64).reshape(-1, 2)
    nrows = len(pp)
    tot = int(np.abs(pp[:, 1]).sum()) if nrows else 0
    cap_bits = (tot // max(samples_per_symbol, 1) + 2 * nrows + 2) * bits_per_symbol
    cap_msg = nrows + 1
    cap_pos = cap_bits + 2 * cap_msg + 2
    bits = np.zeros(cap_bits, dtype=np.uint8)
    msg_off = np.zeros(cap_msg + 1, dtype=np.int64)
    pauses = np.zeros(cap_msg, dtype=np.int64)
    pos = np.zeros(cap_pos if write_bit_sample_pos else 1, dtype=np.int64)
    pos_off = np.zeros(cap_msg + 1, dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    nmsg = lib().orc_ppseq_to_bits(vp(pp), C.c_int64(nrows), C.c_int64(samples_per_symbol), int(bits_per_symbol),
                                   int(bool(write_bit_sample_pos)), C.c_int64(pause_threshold),
                                   vp(bits), C.c_int64(cap_bits), vp(msg_off), vp(pauses), C.c_int64(cap_msg),
                                   vp(pos), C.c_int64(cap_pos), vp(pos_off))
    if nmsg < 0:
        raise RuntimeError("oracle ppseq_to_bits capacity overflow")
    return (bits[:msg_off[nmsg]].copy(), msg_off[:nmsg + 1].copy(), pauses[:nmsg].copy(),
            pos[:pos_off[nmsg]].copy() if write_bit_sample_pos else pos[:0].copy(), pos_off[:nmsg + 1].copy())


def ppseq_to_bits(ppseq, samples_per_symbol: int, bits_per_symbol: int, write_bit_sample_pos=True,
                  pause_threshold=8):
    """Same return shape as the reference: (list of array('B'), array('L') pauses, list of array('L'))."""
    bits, off, pauses, pos, poff = ppseq_to_bits_flat(ppseq, samples_per_symbol, bits_per_symbol,
                                                      write_bit_sample_pos, pause_threshold)
    data = [array.array("B", bits[off[i]:off[i + 1]].tolist()) for i in range(len(pauses))]
    pa = array.array("L", [int(p) for p in pauses])
    bsp = [array.array("L", pos[poff[i]:poff[i + 1]].tolist()) for i in range(len(pauses))] \
        if write_bit_sample_pos else []
    return data, pa, bsp


# ------------------------------------------------------------------------------------------------
# numpy restatements of the Python-side estimators (AutoInterpretation.py)
# ------------------------------------------------------------------------------------------------
def minmax(arr):
    """util.pyx:20-36.  The Cython function returns its C `float` results as Python floats (doubles): detect_center's
    bin edges np.arange(min, max + step, step) are therefore float64 -- returning numpy float32 scalars here would
    make them float32 (NEP 50) and shift the detected center in the 6th digit."""
    if len(arr) == 0:
        return 0, 0
    return float(arr.min()), float(arr.max())


def detect_noise_level(magnitudes) -> float:
    """AutoInterpretation.py:60-91 (chunk means in numpy's own pairwise float64 summation)."""
    n = len(magnitudes)
    if n <= 3:
        return 0
    chunksize = max(1, int(n * 1 / 100))
    ends = [i for i in range(n, 0, -chunksize) if i - chunksize >= 0]
    chunks = [magnitudes[i - chunksize:i] for i in ends]
    mean_values = np.fromiter((np.mean(c) for c in chunks), dtype=np.float32, count=len(chunks))
    minimum, maximum = minmax(mean_values)
    if maximum == 0 or minimum / maximum > 0.9:
        return 0
    indices = np.nonzero(mean_values <= 1.1 * np.min(mean_values))[0]
    try:
        result = np.max([np.max(chunks[i]) for i in indices if len(chunks[i]) > 0])
    except ValueError:
        return 0
    return math.ceil(result * 10000) / 10000


def detect_center(rectangular_signal: np.ndarray, max_size=None):
    """AutoInterpretation.py:226-277"""
    rect = rectangular_signal[rectangular_signal > -4]
    rect = rect[int(0.05 * len(rect)):int(0.95 * len(rect))]
    if max_size is not None and len(rect) > max_size:
        rect = rect[0:max_size]
    hist_min, hist_max = minmax(rect)
    hist_step = float(np.var(rect))
    try:
        y, x = np.histogram(rect, bins=np.arange(hist_min, hist_max + hist_step, hist_step))
    except (ZeroDivisionError, ValueError):
        return None
    num_values = 2
    most_common_levels = []
    window_size = max(2, int(0.05 * len(y)) + 1)

    def get_elem(arr, index, default):
        return arr[index] if 0 <= index < len(arr) else default

    for index in np.argsort(y)[::-1]:
        if all(y[index] > get_elem(y, index + i, 0) and y[index] > get_elem(y, index - i, 0)
               for i in range(1, window_size)):
            most_common_levels.append(x[index])
        if len(most_common_levels) == num_values:
            break
    if len(most_common_levels) == 0:
        return None
    return np.mean(most_common_levels)


def segment_messages_from_magnitudes(magnitudes, noise_threshold: float):
    """auto_interpretation.pyx:55-111 (pure-Python loop: small inputs only)."""
    result = []
    N = len(magnitudes)
    if N == 0:
        return []
    nt = np.float32(noise_threshold)
    start = 0
    outlier_tolerance = 10
    conseq_above = conseq_below = 0
    state = 1 if magnitudes[0] > nt else -1
    for i in range(N):
        above = magnitudes[i] > nt
        if state == 1:
            conseq_below = 0 if above else conseq_below + 1
        elif state == -1:
            conseq_above = conseq_above + 1 if above else 0
        if state == 1 and conseq_below >= outlier_tolerance:
            state = -1
            result.append((start, i - conseq_below))
            conseq_below = conseq_above = 0
        elif state == -1 and conseq_above >= outlier_tolerance:
            state = 1
            start = i - conseq_above
            conseq_below = conseq_above = 0
    if state == 1 and start < N - conseq_below:
        result.append((start, N - conseq_below))
    return result


def merge_message_segments_for_ook(segments: list):
    """AutoInterpretation.merge_message_segments_for_ook (AutoInterpretation.py:107-148), with min_without_outliers(z=1) (:21-25)
    inlined: OOK pulses whose separating pause is shorter than 8 x the minimum pulse within one standard deviation of the mean
    belong to the same message."""
    if len(segments) <= 1:
        return segments
    pulses = np.array([b - a for a, b in segments], dtype=np.uint64)
    pauses = np.array([segments[i + 1][0] - segments[i][1] for i in range(len(segments) - 1)], dtype=np.uint64)
    inliers = pulses[abs(pulses - np.mean(pulses)) <= 1 * np.std(pulses)]
    long_pause = pauses >= 8 * np.min(inliers)
    result = []
    first = 0
    for i in range(len(segments)):
        if i == len(segments) - 1 or long_pause[i]:                    # message = segments first..i
            length = sum(segments[j][1] - segments[j][0] for j in range(first, i + 1))
            length += sum(segments[j][0] - segments[j - 1][1] for j in range(first + 1, i + 1))
            result.append((segments[first][0], segments[first][0] + length))
            first = i + 1
    return result


def get_plateau_lengths(rect_data, center, percentage=25) -> np.ndarray:
    """auto_interpretation.pyx:179-208 (pure-Python loop: small inputs only)."""
    n = len(rect_data)
    if n == 0 or center is None:
        return np.array([], dtype=np.uint64)
    c = np.float32(center)
    state = -1 if rect_data[0] <= c else 1
    plateau_length = 0
    current_sum = 0
    result = []
    for i in range(n):
        if current_sum >= (percentage * n) // 100:      # cdivision: C integer division
            break
        new_state = -1 if rect_data[i] <= c else 1
        if state == new_state:
            plateau_length += 1
        else:
            result.append(plateau_length)
            current_sum += plateau_length
            state = new_state
            plateau_length = 1
    return np.array(result, dtype=np.uint64)


def modulate_c(bits, samples_per_symbol: int, modulation_type: str, parameters, bits_per_symbol: int,
               carrier_amplitude: float, carrier_frequency: float, carrier_phase: float, sample_rate: float,
               pause: int, start: int, dtype=np.float32, gauss_bt: float = 0.5, filter_width: float = 1.0) -> np.ndarray:
    """signal_functions.modulate_c (signal_functions.pyx:56-177) for ASK / FSK / PSK / OQPSK."""
    mod = modulation_type.upper()
    codes = dict(MOD_CODES, OQPSK=4)
    if mod not in codes:
        raise NotImplementedError(modulation_type)
    if mod == "OQPSK":
        assert bits_per_symbol == 2
    dt = np.dtype(dtype)
    if dt not in (np.dtype(np.int8), np.dtype(np.int16), np.dtype(np.float32)):
        raise ValueError("Unsupported dtype for modulation {}".format(dtype))
    b = np.ascontiguousarray(np.frombuffer(bytes(bytearray(bits)), dtype=np.uint8) if not isinstance(bits, np.ndarray) else bits, dtype=np.uint8)
    par = np.ascontiguousarray(parameters, dtype=np.float32)
    total = (len(b) // bits_per_symbol) * samples_per_symbol + pause
    out = np.zeros((total, 2), dtype=dt)
    k = lib().orc_modulate(b.ctypes.data_as(C.c_void_p), C.c_int64(len(b)), C.c_uint32(samples_per_symbol), codes[mod],
                           par.ctypes.data_as(C.c_void_p), C.c_int(bits_per_symbol), C.c_float(carrier_amplitude),
                           C.c_float(carrier_frequency), C.c_float(carrier_phase), C.c_float(sample_rate), C.c_uint32(pause),
                           C.c_uint32(start), DT_CODES[dt], out.ctypes.data_as(C.c_void_p))
    assert k == total
    return out


def gauss_fir(sample_rate: float, samples_per_symbol: int, bt: float = 0.5, filter_width: float = 1.0) -> np.ndarray:
    """signal_functions.gauss_fir (signal_functions.pyx:230-243): the same numpy expression (sample_rate, bt, filter_width
    arrive as C floats there, i.e. as Python floats holding float32 values)."""
    bt, filter_width, sample_rate = float(np.float32(bt)), float(np.float32(filter_width)), float(np.float32(sample_rate))
    k = np.arange(-int(filter_width * samples_per_symbol), int(filter_width * samples_per_symbol) + 1, dtype=np.float32)
    ts = float(np.float32(samples_per_symbol / sample_rate))          # cdef float ts
    h = (np.sqrt((2 * np.pi) / (np.log(2))) * bt / ts * np.exp(
        -(((np.sqrt(2) * np.pi) / np.sqrt(np.log(2)) * bt * k / samples_per_symbol) ** 2))).astype(np.float32)
    return h / h.sum()


def modulate_gfsk(bits, samples_per_symbol: int, parameters, bits_per_symbol: int, carrier_amplitude: float,
                  carrier_phase: float, sample_rate: float, pause: int, start: int, dtype=np.float32, gauss_bt: float = 0.5,
                  filter_width: float = 1.0, return_freqs_phases: bool = False, frequencies=None):
    """modulate_c(..., "GFSK", ...) (signal_functions.pyx:118-125, 156-163, 196-228); see orc_modulate_gfsk for what is and
    what is not bit-reproducible in the reference.  frequencies: the filtered frequencies to use instead of the restated
    convolution (numpy's, for the reference's bits on this host)."""
    dt = np.dtype(dtype)
    b = np.ascontiguousarray(np.frombuffer(bytes(bytearray(bits)), dtype=np.uint8) if not isinstance(bits, np.ndarray) else bits, dtype=np.uint8)
    par = np.ascontiguousarray(parameters, dtype=np.float32)
    g = np.ascontiguousarray(gauss_fir(sample_rate, samples_per_symbol, gauss_bt, filter_width), dtype=np.float32)
    n = (len(b) // bits_per_symbol) * samples_per_symbol
    out = np.zeros((n + pause, 2), dtype=dt)
    fr, ph = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    fin = None if frequencies is None else np.ascontiguousarray(frequencies, dtype=np.float32)
    assert fin is None or len(fin) == n
    f = lib().orc_modulate_gfsk
    f.restype = C.c_int64
    k = f(b.ctypes.data_as(C.c_void_p), C.c_int64(len(b)), C.c_uint32(samples_per_symbol), par.ctypes.data_as(C.c_void_p),
          C.c_int(bits_per_symbol), C.c_float(carrier_amplitude), C.c_float(carrier_phase), C.c_float(sample_rate), C.c_uint32(pause),
          C.c_uint32(start), g.ctypes.data_as(C.c_void_p), C.c_int64(len(g)),
          fin.ctypes.data_as(C.c_void_p) if fin is not None else None, DT_CODES[dt], out.ctypes.data_as(C.c_void_p),
          fr.ctypes.data_as(C.c_void_p), ph.ctypes.data_as(C.c_void_p))
    if k == -3:
        raise ZeroDivisionError("integer division or modulo by zero")
    assert k == n + pause
    return (out, fr[:n], ph[:n]) if return_freqs_phases else out


def create_path_arrays(samples, start: int, end: int, subpath_ranges=None, pixels_on_path: int = 5000):
    """path_creator.create_path (path_creator.pyx:19-82) up to the point where it hands (x, values) slices to
    array_to_QPath: per-pixel minimum / maximum of samples[start:end] (:46-66), or the samples themselves when there is
    at most one sample per pixel (:67-70), cut into the requested sub-paths (:76-81).  Returns [(x int64, values)]."""
    samples = np.ascontiguousarray(samples)
    num_samples = end - start
    subpath_ranges = [(start, end)] if subpath_ranges is None else subpath_ranges
    spp = (abs(num_samples) // pixels_on_path) * (1 if num_samples >= 0 else -1)   # C division of two long long (:38)
    if spp > 1:
        rng = np.arange(start, end, spp, dtype=np.int64)
        values = np.zeros(2 * len(rng), dtype=samples.dtype)
        scale_factor = float(np.float32(num_samples / (2.0 * len(rng))))      # cdef float
        for k, i in enumerate(rng):
            chunk = samples[i:min(i + spp, end)]
            mn = mx = chunk[0]
            for v in chunk[1:]:                      # :56-61 (a NaN never wins a comparison)
                if v < mn:
                    mn = v
                elif v > mx:
                    mx = v
            values[2 * k], values[2 * k + 1] = mn, mx
        x = np.repeat(rng, 2)
    else:
        x = np.arange(start, end, dtype=np.int64)
        values = samples[start:end]
        scale_factor = 1.0
    if scale_factor == 0:
        scale_factor = 1
    out = []
    for r in subpath_ranges:
        s0 = ((((r[0] - start) / scale_factor) * scale_factor) - 2 * scale_factor) / scale_factor
        s0 = int(max(0, math.floor(s0)))
        s1 = ((((r[1] - start) / scale_factor) * scale_factor) + 2 * scale_factor) / scale_factor
        s1 = int(max(0, math.ceil(s1)))
        out.append((x[s0:s1], values[s0:s1]))
    return out


def convert_to(data: np.ndarray, target_dtype) -> np.ndarray:
    """IQArray.convert_to (IQArray.py:127-203), the numpy expressions of the reference one to one."""
    target_dtype = np.dtype(target_dtype).type
    if target_dtype == data.dtype:
        return data
    if data.dtype == np.uint8:
        if target_dtype == np.int8:
            return np.add(data, -128, dtype=np.int8, casting="unsafe")
        elif target_dtype == np.int16:
            return np.add(data, -128, dtype=np.int16, casting="unsafe") << 8
        elif target_dtype == np.uint16:
            return data.astype(np.uint16) << 8
        elif target_dtype == np.float32:
            return np.add(np.multiply(data, 1 / 128, dtype=np.float32), -1.0, dtype=np.float32)
    if data.dtype == np.int8:
        if target_dtype == np.uint8:
            return np.add(data, 128, dtype=np.uint8, casting="unsafe")
        elif target_dtype == np.int16:
            return data.astype(np.int16) << 8
        elif target_dtype == np.uint16:
            return np.add(data, 128, dtype=np.uint16, casting="unsafe") << 8
        elif target_dtype == np.float32:
            return np.multiply(data, 1 / 128, dtype=np.float32)
    if data.dtype == np.uint16:
        if target_dtype == np.int8:
            return (np.add(data, -32768, dtype=np.int16, casting="unsafe") >> 8).astype(np.int8)
        elif target_dtype == np.uint8:
            return (data >> 8).astype(np.uint8)
        elif target_dtype == np.int16:
            return np.add(data, -32768, dtype=np.int16, casting="unsafe")
        elif target_dtype == np.float32:
            return np.add(np.multiply(data, 1 / 32768, dtype=np.float32), -1.0, dtype=np.float32)
    if data.dtype == np.int16:
        if target_dtype == np.int8:
            return (data >> 8).astype(np.int8)
        elif target_dtype == np.uint8:
            return (np.add(data, 32768, dtype=np.uint16, casting="unsafe") >> 8).astype(np.uint8)
        elif target_dtype == np.uint16:
            return np.add(data, 32768, dtype=np.uint16, casting="unsafe")
        elif target_dtype == np.float32:
            return np.multiply(data, 1 / 32768, dtype=np.float32)
    if data.dtype == np.float32:
        if target_dtype == np.int8:
            return np.multiply(data, 127, dtype=np.float32).astype(np.int8)
        elif target_dtype == np.uint8:
            return np.multiply(np.add(data, 1.0, dtype=np.float32), 127, dtype=np.float32).astype(np.uint8)
        elif target_dtype == np.int16:
            return np.multiply(data, 32767, dtype=np.float32).astype(np.int16)
        elif target_dtype == np.uint16:
            return np.multiply(np.add(data, 1.0, dtype=np.float32), 32767, dtype=np.float32).astype(np.uint16)
    raise ValueError("Data type {} not supported".format(target_dtype))
