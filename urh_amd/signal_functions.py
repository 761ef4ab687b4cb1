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
    return C.c_void_p(a.ctypes.data)


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


def _bits_u8(bits) -> np.ndarray:
    if isinstance(bits, np.ndarray):
        return np.ascontiguousarray(bits, dtype=np.uint8)
    if isinstance(bits, str):
        return np.frombuffer(bytes(map(int, bits)), dtype=np.uint8).copy()
    return np.frombuffer(bytes(bytearray(bits)), dtype=np.uint8).copy()


_MOD_DTYPES = {np.dtype(np.int8): _lib.DT_I8, np.dtype(np.int16): _lib.DT_I16, np.dtype(np.float32): _lib.DT_F32}


def _modulation_code(modulation_type: str) -> int:
    m = modulation_type.lower()                      # the reference compares lower-case names (:111-115)
    if m in ("ask", "fsk", "psk"):
        return _MOD[m.upper()]
    if m == "oqpsk":
        return 4                                     # URHGPU_MOD_OQPSK
    if m == "gfsk":
        return 5                                     # urhgpu_modulate_gfsk*
    raise AssertionError(modulation_type)            # `assert is_fsk or is_ask or ...` (:117)


def gauss_fir(sample_rate: float, samples_per_symbol: int, bt: float = 0.5, filter_width: float = 1.0) -> np.ndarray:
    """signal_functions.gauss_fir (signal_functions.pyx:230-243): the Gaussian taps, evaluated with numpy on the host as
    the reference does (its arguments are C floats there: Python floats holding float32 values)."""
    bt, filter_width, sample_rate = float(np.float32(bt)), float(np.float32(filter_width)), float(np.float32(sample_rate))
    half = int(filter_width * samples_per_symbol)
    k = np.arange(-half, half + 1, dtype=np.float32)
    ts = float(np.float32(samples_per_symbol / sample_rate))                               # symbol time (cdef float)
    h = (np.sqrt((2 * np.pi) / (np.log(2))) * bt / ts * np.exp(
        -(((np.sqrt(2) * np.pi) / np.sqrt(np.log(2)) * bt * k / samples_per_symbol) ** 2))).astype(np.float32)
    return h / h.sum()


def gfsk_frequencies_numpy(bits, parameters, samples_per_symbol: int, bits_per_symbol: int, gfir) -> np.ndarray:
    """The Gaussian-filtered per-sample frequencies as the reference computes them ON THIS HOST: numpy's float32
    convolution (signal_functions.pyx:203-217).  Pass the result as `gfsk_frequencies` to modulate_c to get the reference's
    bits (numpy's BLAS dot product decides the last bit of every frequency, and the phase recurrence amplifies it)."""
    b = _bits_u8(bits)
    n_sym = len(b) // int(bits_per_symbol)
    par = np.ascontiguousarray(parameters, dtype=np.float32)
    sym = b[:n_sym * bits_per_symbol].reshape(n_sym, bits_per_symbol).astype(np.int64)
    index = (sym << np.arange(bits_per_symbol - 1, -1, -1)).sum(axis=1)
    raw = np.repeat(par[index], int(samples_per_symbol)).astype(np.float32)
    gfir = np.ascontiguousarray(gfir, dtype=np.float32)
    if len(raw) >= len(gfir):
        return np.convolve(raw, gfir, mode="same")
    return np.convolve(gfir, raw, mode="same")[:len(raw)]


def _modulate_gfsk(b, samples_per_symbol, par, bits_per_symbol, carrier_amplitude, carrier_phase, sample_rate, pause, start, dt,
                   gauss_bt, filter_width, gfsk_frequencies, out, ctx):
    if len(b) // int(bits_per_symbol) == 0:
        raise ZeroDivisionError("integer division or modulo by zero")                     # len(bits) // num_symbols (:201)
    gfir = np.ascontiguousarray(gauss_fir(sample_rate, samples_per_symbol, gauss_bt, filter_width), dtype=np.float32)
    freqs = None
    if isinstance(gfsk_frequencies, str):
        if gfsk_frequencies != "numpy":
            raise ValueError("gfsk_frequencies: None (device convolution), 'numpy' or an array")
        freqs = gfsk_frequencies_numpy(b, par, samples_per_symbol, bits_per_symbol, gfir)
    elif gfsk_frequencies is not None:
        freqs = gfsk_frequencies
    if freqs is not None:
        freqs = np.ascontiguousarray(freqs, dtype=np.float32)
        if len(freqs) != (len(b) // int(bits_per_symbol)) * int(samples_per_symbol):
            raise ValueError("gfsk_frequencies: one value per data sample expected")
    _lib.check(_lib.load().urhgpu_modulate_gfsk(ctx.handle, _vp(b), len(b), int(samples_per_symbol), _vp(par), int(bits_per_symbol),
                                                float(carrier_amplitude), float(carrier_phase), float(sample_rate), int(pause),
                                                int(start), _MOD_DTYPES[dt], _vp(gfir), len(gfir),
                                                _vp(freqs) if freqs is not None else None, _vp(out)))
    return out


def modulate_c(bits, samples_per_symbol: int, modulation_type: str, parameters, bits_per_symbol: int,
               carrier_amplitude: float, carrier_frequency: float, carrier_phase: float, sample_rate: float,
               pause: int, start: int, dtype=np.float32, gauss_bt: float = 0.5, filter_width: float = 1.0,
               ctx=None, gfsk_frequencies=None) -> np.ndarray:
    """signal_functions.modulate_c (signal_functions.pyx:56-177): bits -> (N, 2) IQ samples of `dtype`
    (np.float32 / np.int8 / np.int16), N = len(bits) // bits_per_symbol * samples_per_symbol + pause.
    GFSK: gfsk_frequencies=None convolves on the GPU (exactly accumulated dot products), "numpy" takes the Gaussian-filtered
    frequencies from numpy's convolution on the host like the reference (see gfsk_frequencies_numpy)."""
    dt = np.dtype(dtype)
    if dt not in _MOD_DTYPES:
        raise ValueError("Unsupported dtype for modulation {}".format(dtype))
    b = _bits_u8(bits)
    total = (len(b) // int(bits_per_symbol)) * int(samples_per_symbol) + int(pause)
    out = np.zeros((total, 2), dtype=dt)
    if len(b) == 0:
        return out                                   # :104-106
    mod = _modulation_code(modulation_type)
    if mod == 5 and len(b) // int(bits_per_symbol) == 0:
        raise ZeroDivisionError("integer division or modulo by zero")                     # len(bits) // num_symbols (:201)
    if total == 0:
        return out
    assert mod != 4 or int(bits_per_symbol) == 2     # :120
    par = np.ascontiguousarray(parameters, dtype=np.float32)
    if len(par) < (1 << int(bits_per_symbol)):
        raise IndexError("parameters shorter than 2**bits_per_symbol")
    ctx = ctx or _lib.default_context()
    if mod == 5:
        return _modulate_gfsk(b, samples_per_symbol, par, bits_per_symbol, carrier_amplitude, carrier_phase, sample_rate, pause,
                              start, dt, gauss_bt, filter_width, gfsk_frequencies, out, ctx)
    _lib.check(_lib.load().urhgpu_modulate(ctx.handle, _vp(b), len(b), int(samples_per_symbol), mod, _vp(par),
                                           int(bits_per_symbol), float(carrier_amplitude), float(carrier_frequency),
                                           float(carrier_phase), float(sample_rate), int(pause), int(start),
                                           _MOD_DTYPES[dt], _vp(out)))
    return out


def modulate_messages_dev(messages, samples_per_symbol: int, modulation_type: str, parameters, bits_per_symbol: int,
                          carrier_amplitude: float, carrier_frequency: float, carrier_phase: float, sample_rate: float,
                          pauses, starts=None, dtype=np.float32, device=None, ctx=None, gauss_bt: float = 0.5,
                          filter_width: float = 1.0):
    """Many messages rendered back to back by one launch, result left in HBM: a torch tensor (N, 2) of `dtype`.
    messages: sequence of bit sequences; pauses[m] silent samples follow message m; starts[m] is the sample index of
    its first sample (default: consecutive, as ProtocolAnalyzerContainer.modulate places them)."""
    import torch
    dt = np.dtype(dtype)
    if dt not in _MOD_DTYPES:
        raise ValueError("Unsupported dtype for modulation {}".format(dtype))
    mod = _modulation_code(modulation_type)
    assert mod != 4 or int(bits_per_symbol) == 2     # :120
    bl = [_bits_u8(m) for m in messages]
    off = np.zeros(len(bl) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in bl])
    allbits = np.concatenate(bl) if bl else np.zeros(0, np.uint8)
    pa = np.ascontiguousarray(pauses, dtype=np.uint32)
    lens = np.array([(len(x) // int(bits_per_symbol)) * int(samples_per_symbol) for x in bl], dtype=np.int64) + pa
    if starts is None:
        st = np.zeros(len(bl), dtype=np.int64)
        st[1:] = np.cumsum(lens)[:-1]
    else:
        st = np.asarray(starts, dtype=np.int64)
    st = np.ascontiguousarray(st, dtype=np.uint32)
    total = int(lens.sum())
    par = np.ascontiguousarray(parameters, dtype=np.float32)
    ctx = ctx or _lib.default_context()
    tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16}[dt]
    out = torch.empty((total, 2), dtype=tdt, device=device if device is not None else torch.device("cuda", ctx.device))
    ctx.set_stream(torch.cuda.current_stream(out.device).cuda_stream)
    got = C.c_int64(0)
    if mod == 5:
        gfir = np.ascontiguousarray(gauss_fir(sample_rate, samples_per_symbol, gauss_bt, filter_width), dtype=np.float32)
        _lib.check(_lib.load().urhgpu_modulate_gfsk_dev(ctx.handle, _vp(allbits), _vp(off), _vp(pa), _vp(st), len(bl),
                                                        int(samples_per_symbol), _vp(par), int(bits_per_symbol),
                                                        float(carrier_amplitude), float(carrier_phase), float(sample_rate),
                                                        _MOD_DTYPES[dt], _vp(gfir), len(gfir), None, C.c_void_p(out.data_ptr()),
                                                        total, C.byref(got)))
        assert got.value == total
        return out
    _lib.check(_lib.load().urhgpu_modulate_dev(ctx.handle, _vp(allbits), _vp(off), _vp(pa), _vp(st), len(bl),
                                               int(samples_per_symbol), mod, _vp(par), int(bits_per_symbol),
                                               float(carrier_amplitude), float(carrier_frequency), float(carrier_phase),
                                               float(sample_rate), _MOD_DTYPES[dt], C.c_void_p(out.data_ptr()), total,
                                               C.byref(got)))
    assert got.value == total
    return out
