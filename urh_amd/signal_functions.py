"""Drop-in mirror of the reference's `urh.cythonext.signal_functions` (the functions on the IQ->bits
path), backed by the HIP kernels in liburhgpu.so through the C ABI (include/urhgpu.h).

Same names, argument meaning, return types and edge cases as
/root/reference/src/urh/cythonext/signal_functions.pyx:
    afp_demod :333-378, get_center_thresholds :380-390, grab_pulse_lens :392-495,
    fir_filter :513-525, iir_filter :527-542
plus `ppseq_to_bits`, the device version of the pure-Python tail
ProtocolAnalyzer._ppseq_to_bits (/root/reference/src/urh/signalprocessing/ProtocolAnalyzer.py:323-414).

There is no CPU fallback: without liburhgpu.so or without a GPU every call raises.
"""
import array
import ctypes as C

import numpy as np

from . import _lib

_DT = {np.dtype(np.int8): _lib.DT_I8, np.dtype(np.uint8): _lib.DT_U8, np.dtype(np.int16): _lib.DT_I16,
       np.dtype(np.uint16): _lib.DT_U16, np.dtype(np.float32): _lib.DT_F32}
_MOD = {"ASK": _lib.MOD_ASK, "FSK": _lib.MOD_FSK, "PSK": _lib.MOD_PSK}


def noise_for_mod_type(mod_type: str) -> float:
    """get_noise_for_mod_type (signal_functions.pyx:31-44)"""
    if mod_type == "ASK":
        return 0.0
    if mod_type in ("FSK", "PSK", "OQPSK"):
        return -4.0
    if mod_type == "QAM":
        return -0.0
    return 0.0


def mod_code(mod_type: str):
    """(URHGPU_MOD_* code, NOISE sentinel for MOD_OTHER)"""
    if mod_type in _MOD:
        return _MOD[mod_type], 0.0
    return _lib.MOD_OTHER, noise_for_mod_type(mod_type)


def dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt not in _DT:
        raise ValueError("Unsupported dtype")
    return _DT[dt]


def _iq(samples) -> np.ndarray:
    """IQ samples as a C-contiguous (N, 2) array of a supported dtype (util.pxd:1-8)."""
    a = samples
    if not isinstance(a, np.ndarray):
        a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected (N, 2))")
    if a.dtype not in _DT:
        raise ValueError("Unsupported dtype")
    return np.ascontiguousarray(a)


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def afp_demod(samples, noise_mag: float, mod_type: str, mod_order: int, costas_loop_bandwidth: float = 0.1,
              ctx=None) -> np.ndarray:
    a = _iq(samples)
    n = len(a)
    out = np.zeros(n, dtype=np.float32)
    if n <= 2:                                   # signal_functions.pyx:335-336
        return out
    ctx = ctx or _lib.default_context()
    mod, sentinel = mod_code(mod_type)
    _lib.check(_lib.load().urhgpu_afp_demod(ctx.handle, _vp(a), _DT[a.dtype], n, float(noise_mag), mod, int(mod_order),
                                            float(costas_loop_bandwidth), float(sentinel), _vp(out)))
    return out


def get_center_thresholds(center: float, spacing: float, modulation_order: int) -> np.ndarray:
    out = np.empty(max(int(modulation_order) - 1, 0), dtype=np.float32)
    _lib.check(_lib.load().urhgpu_get_center_thresholds(float(center), float(spacing), int(modulation_order), _vp(out)))
    return out


def grab_pulse_lens(samples, center: float, tolerance: int, modulation_type: str, samples_per_symbol: int,
                    bits_per_symbol: int = 1, center_spacing: float = 0.1, ctx=None) -> np.ndarray:
    """Pulse table int64[P, 2]: rows [state, length], state -1 = pause."""
    s = np.ascontiguousarray(samples, dtype=np.float32)
    if s.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1)")
    n = len(s)
    if n == 0:                                   # signal_functions.pyx:416-417
        return np.zeros((0, 2), dtype=np.int64)
    if not 0 <= int(tolerance) <= 0xFFFF:
        raise OverflowError("value too large to convert to uint16_t")
    ctx = ctx or _lib.default_context()
    mod, sentinel = mod_code(modulation_type)
    cap = n // (int(tolerance) + 1) + 2
    cap = min(cap, max(1024, n // 8 + 2))        # usually plenty; retried below if not
    n_rows = C.c_int64(0)
    while True:
        rows = np.zeros((cap, 2), dtype=np.int64)
        st = _lib.load().urhgpu_grab_pulse_lens(ctx.handle, _vp(s), n, float(center), int(tolerance), mod,
                                                int(samples_per_symbol), int(bits_per_symbol), float(center_spacing),
                                                float(sentinel), _vp(rows), cap, C.byref(n_rows))
        if st == _lib.ERR_CAPACITY:
            cap = int(n_rows.value)
            continue
        _lib.check(st)
        return rows[:n_rows.value]


def ppseq_to_bits_flat(ppseq, samples_per_symbol: int, bits_per_symbol: int, write_bit_sample_pos=True,
                       pause_threshold=8, ctx=None):
    """Flat form: (bits u8[n_bits], msg_off i64[n_msg+1], pauses i64[n_msg], pos i64[n_pos], pos_off i64[n_msg+1])."""
    pp = np.ascontiguousarray(ppseq, dtype=np.int64).reshape(-1, 2)
    nrows = len(pp)
    ctx = ctx or _lib.default_context()
    cap_bits = max(64, nrows * 4 * int(bits_per_symbol))
    cap_msg = max(16, nrows // 8)
    cap_pos = cap_bits + 2 * cap_msg + 2
    counts = np.zeros(4, dtype=np.int64)
    while True:
        bits = np.zeros(cap_bits, dtype=np.uint8)
        msg_off = np.zeros(cap_msg + 1, dtype=np.int64)
        pauses = np.zeros(cap_msg, dtype=np.int64)
        pos = np.zeros(cap_pos if write_bit_sample_pos else 1, dtype=np.int64)
        pos_off = np.zeros(cap_msg + 1, dtype=np.int64)
        st = _lib.load().urhgpu_ppseq_to_bits(ctx.handle, _vp(pp), nrows, int(samples_per_symbol), int(bits_per_symbol),
                                              1 if write_bit_sample_pos else 0, int(pause_threshold),
                                              _vp(bits), cap_bits, _vp(msg_off), _vp(pauses), cap_msg,
                                              _vp(pos), cap_pos if write_bit_sample_pos else 0, _vp(pos_off), _vp(counts))
        if st == _lib.ERR_CAPACITY:
            cap_msg = max(cap_msg, int(counts[0]))
            cap_bits = max(cap_bits, int(counts[1]))
            cap_pos = max(cap_pos, int(counts[2]))
            continue
        _lib.check(st)
        n_msg, n_bits, n_pos = (int(c) for c in counts[:3])
        return (bits[:n_bits], msg_off[:n_msg + 1], pauses[:n_msg],
                pos[:n_pos] if write_bit_sample_pos else pos[:0], pos_off[:n_msg + 1])


def ppseq_to_bits(ppseq, samples_per_symbol: int, bits_per_symbol: int, write_bit_sample_pos=True,
                  pause_threshold=8, ctx=None):
    """Same return value as ProtocolAnalyzer._ppseq_to_bits:
    (list of array('B') bit arrays, array('L') pauses, list of array('L') bit sample positions)."""
    bits, off, pauses, pos, poff = ppseq_to_bits_flat(ppseq, samples_per_symbol, bits_per_symbol,
                                                      write_bit_sample_pos, pause_threshold, ctx)
    data = [array.array("B", bits[off[i]:off[i + 1]].tobytes()) for i in range(len(pauses))]
    pa = array.array("L", pauses.tolist())
    bsp = [array.array("L", pos[poff[i]:poff[i + 1]].tolist()) for i in range(len(pauses))] \
        if write_bit_sample_pos else []
    return data, pa, bsp


def fir_filter(input_samples, filter_taps, ctx=None) -> np.ndarray:
    x = np.ascontiguousarray(input_samples, dtype=np.complex64)
    h = np.ascontiguousarray(filter_taps, dtype=np.complex64)
    out = np.zeros(len(x), dtype=np.complex64)
    if len(x) == 0:
        return out
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_fir_filter(ctx.handle, _vp(x), len(x), _vp(h), len(h), _vp(out)))
    return out


def iir_filter(a, b, signal, ctx=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.ascontiguousarray(signal, dtype=np.complex64)
    out = np.zeros(len(x), dtype=np.complex64)
    if len(x) == 0:
        return out
    ctx = ctx or _lib.default_context()
    _lib.check(_lib.load().urhgpu_iir_filter(ctx.handle, _vp(a), len(a), _vp(b), len(b), _vp(x), len(x), _vp(out)))
    return out
