"""Synthetic capture generator for benchmarks and full-size tests: continuous-phase 2-FSK / OOK with
AWGN, built directly in HBM with torch (plumbing: not part of the IQ->bits path).

The layout follows SURVEY.md §8(d) config 2: the capture is a sequence of 2^20-sample segments, each
`symbols_per_segment` random symbols at `sps` samples/symbol followed by a short silent gap, +-20 kHz
deviation at 1 MS/s, AWGN sigma 0.05.  Generation is seeded and deterministic for a given torch build;
the transmitted bits are returned so that tests can check the recovered bits against them.
"""
import math


def fsk_capture(n_segments: int, device, seed: int = 1234, sps: int = 100, seg_len: int = 1 << 20,
                deviation_hz: float = 20e3, sample_rate: float = 1e6, noise: float = 0.05, first_segment: int = 0):
    """Returns (iq float32 (n_segments*seg_len, 2) on `device`, bits uint8 (n_segments, symbols_per_segment))."""
    import torch
    nsym = seg_len // sps                      # 10485 symbols + 76 silent samples for the defaults
    n = n_segments * seg_len
    iq = torch.empty((n, 2), dtype=torch.float32, device=device)
    all_bits = torch.empty((n_segments, nsym), dtype=torch.uint8, device=device)
    step = 2.0 * math.pi * deviation_hz / sample_rate
    for k in range(n_segments):
        g = torch.Generator(device=device)
        g.manual_seed(seed + first_segment + k)
        bits = torch.randint(0, 2, (nsym,), generator=g, device=device, dtype=torch.int64)
        all_bits[k] = bits.to(torch.uint8)
        inc = (bits.to(torch.float64) * 2.0 - 1.0) * step
        phase = torch.cumsum(inc.repeat_interleave(sps), 0)
        seg = iq[k * seg_len:(k + 1) * seg_len]
        seg.zero_()
        seg[:nsym * sps, 0] = torch.cos(phase).to(torch.float32)
        seg[:nsym * sps, 1] = torch.sin(phase).to(torch.float32)
        seg.add_(torch.randn((seg_len, 2), generator=g, device=device, dtype=torch.float32), alpha=noise)
    return iq, all_bits


SPEC_SEG = 1 << 20


def spec_fsk_bits(k: int, seg_len: int = SPEC_SEG, sps: int = 100, n_symbols=None):
    """Bits of segment k of SURVEY.md §8(d) config 2 / config 4: numpy.random.default_rng(1234 + k).integers(0, 2, seg_len // sps)
    (variant 2b, bursty: n_symbols = 10 465, leaving a 2 076-sample gap per segment)."""
    import numpy as np
    return np.random.default_rng(1234 + k).integers(0, 2, seg_len // sps if n_symbols is None else n_symbols).astype(np.uint8)


def spec_fsk_capture(n_segments: int, device, first_segment: int = 0, sps: int = 100, seg_len: int = SPEC_SEG,
                     deviation_hz: float = 20e3, noise: float = 0.05, host_modulate=None, n_symbols=None):
    """The capture SURVEY.md §8(d) config 2 specifies, byte for byte (config 4: `first_segment` = 128 * rank):

        segment k:  bits_k = numpy.random.default_rng(1234 + k).integers(0, 2, seg_len // sps)
                    modulate_c(bits_k, sps, "FSK", [-20e3, +20e3], 1, 1.0, 40e3, 0, 1e6, pause = seg_len - len(bits_k) * sps, start = 0)
                    + noise * numpy.random.default_rng(5678 + k).standard_normal((seg_len, 2)).astype(float32)

    `modulate_c` is this library's GPU generator (bit-exact against the reference's, tests/test_modulate.py: all segments in one
    launch) unless `host_modulate` is given (a callable with modulate_c's signature, e.g. the real reference's compiled module:
    tests use it to check the bytes).  The AWGN comes from numpy on the host, segment by segment, and is added on the GPU with two
    separately rounded float32 operations (multiply, add), i.e. exactly what numpy computes for `iq + noise * awgn`.
    Returns (iq float32 (n_segments * seg_len, 2) on `device`, bits uint8 (n_segments, seg_len // sps) on the host)."""
    import numpy as np
    import torch
    from .signal_functions import modulate_messages_dev
    nsym = seg_len // sps if n_symbols is None else int(n_symbols)
    pause = seg_len - nsym * sps
    ks = [first_segment + k for k in range(n_segments)]
    bits = np.stack([spec_fsk_bits(k, seg_len, sps, nsym) for k in ks]) if ks else np.zeros((0, nsym), np.uint8)
    par = np.array([-deviation_hz, deviation_hz], dtype=np.float32)
    dev = torch.device(device)
    if host_modulate is None:
        with torch.cuda.device(dev):
            from . import _lib
            iq = modulate_messages_dev(list(bits), sps, "FSK", par, 1, 1.0, 40e3, 0.0, 1e6, [pause] * n_segments,
                                       starts=[0] * n_segments, device=dev, ctx=_lib.Context(dev.index))
            torch.cuda.synchronize(dev)
    else:
        import array
        iq = torch.empty((n_segments * seg_len, 2), dtype=torch.float32, device=dev)
        for j in range(n_segments):
            seg = np.asarray(host_modulate(array.array("B", bits[j].tolist()), sps, "FSK", array.array("f", par.tolist()), 1, 1.0, 40e3, 0.0,
                                           1e6, pause, 0))
            iq[j * seg_len:(j + 1) * seg_len] = torch.from_numpy(np.ascontiguousarray(seg, dtype=np.float32)).to(dev)
    assert iq.shape[0] == n_segments * seg_len
    if noise:
        for j, k in enumerate(ks):
            awgn = np.random.default_rng(5678 + k).standard_normal((seg_len, 2)).astype(np.float32)
            w = torch.from_numpy(awgn).to(dev)
            w.mul_(noise)                                   # float32(noise) * awgn, rounded
            iq[j * seg_len:(j + 1) * seg_len].add_(w)       # one more rounding: no fused multiply-add across the two
    return iq, bits


def spec_fir_taps():
    """the first 64 taps of Filter.design_windowed_sinc_bandpass(f_low=0.02, f_high=0.06,
    bw=4/64), cast to complex64 (a 64-tap filter enters the reference as custom taps -> Filter.apply_fir_filter -> fir_filter)."""
    import numpy as np
    from .filter import design_windowed_sinc_bandpass
    return np.ascontiguousarray(design_windowed_sinc_bandpass(0.02, 0.06, 4 / 64)[:64], dtype=np.complex64)


def _spec_segments(n_segments, device, first_segment, seg_len, messages, pauses, modulation, par, bps, awgn, noise_segments=()):
    """common part of the config 3 / config 5 generators: all segments through ONE modulate launch (start = 0 per segment), then
    `awgn(k)` (float32 (seg_len, 2), host numpy) added per segment with one float32 rounding"""
    import numpy as np
    import torch
    from . import _lib
    from .signal_functions import modulate_messages_dev
    dev = torch.device(device)
    live = [j for j in range(n_segments) if (first_segment + j) not in noise_segments]
    iq = torch.zeros((n_segments * seg_len, 2), dtype=torch.float32, device=dev)
    if live:
        with torch.cuda.device(dev):
            body = modulate_messages_dev([messages[j] for j in live], 100, modulation, par, bps, 1.0, 40e3, 0.0, 1e6,
                                         [pauses[j] for j in live], starts=[0] * len(live), device=dev, ctx=_lib.Context(dev.index))
            torch.cuda.synchronize(dev)
        assert body.shape[0] == len(live) * seg_len
        for i, j in enumerate(live):
            iq[j * seg_len:(j + 1) * seg_len] = body[i * seg_len:(i + 1) * seg_len]
        del body
    for j in range(n_segments):
        iq[j * seg_len:(j + 1) * seg_len].add_(torch.from_numpy(awgn(first_segment + j)).to(dev))
    return iq


def spec_ook_capture(n_segments: int, device, first_segment: int = 0, seg_len: int = SPEC_SEG):
    """segments 0-3 noise only; segment k >= 4: Manchester expansion (1 -> 10, 0 -> 01) of
    default_rng(4321 + k).integers(0, 2, 5000) = 10 000 chips, modulate_c(chips, 100, "ASK", [0.0, 1.0], 1, 1.0, 40e3, 0, 1e6,
    pause = 48 576, start = 0); AWGN 0.02 * default_rng(8765 + k).standard_normal((seg_len, 2)).astype(float32).
    Returns (iq float32 (n, 2) on device, chips uint8 (n_segments, 10000) on the host; noise-only segments: zeros)."""
    import numpy as np
    chips = np.zeros((n_segments, 10000), np.uint8)
    msgs, pauses = [], []
    for j in range(n_segments):
        k = first_segment + j
        b = np.random.default_rng(4321 + k).integers(0, 2, 5000).astype(np.uint8)
        c = np.empty(10000, np.uint8)
        c[0::2], c[1::2] = b, 1 - b
        if k >= 4:
            chips[j] = c
        msgs.append(c)
        pauses.append(seg_len - 10000 * 100)
    par = np.array([0.0, 1.0], dtype=np.float32)
    awgn = lambda k: np.float32(0.02) * np.random.default_rng(8765 + k).standard_normal((seg_len, 2)).astype(np.float32)   # noqa: E731
    iq = _spec_segments(n_segments, device, first_segment, seg_len, msgs, pauses, "ASK", par, 1, awgn, noise_segments=(0, 1, 2, 3))
    return iq, chips


def spec_psk_capture(n_segments: int, device, first_segment: int = 0, seg_len: int = SPEC_SEG):
    """segment k: default_rng(2468 + k).integers(0, 2, 20 970) (2 bits/symbol, 10 485 symbols),
    modulate_c(bits, 100, "PSK", [-135, -45, 45, 135] deg in radians, 2, 1.0, 40e3, 0, 1e6, pause = 76, start = 0) as in
    tests/test_demodulations.py:90-94; AWGN 0.1 * N(0, sqrt(2)/2) per component (:109-112) from default_rng(42 + k), added in
    float64 and rounded to float32 once, as the reference test does (`noised.astype(np.float32)`)."""
    import numpy as np
    import torch
    bits = np.stack([np.random.default_rng(2468 + first_segment + j).integers(0, 2, 20970).astype(np.uint8) for j in range(n_segments)])
    par = np.array([np.pi * a / 180 for a in (-135, -45, 45, 135)], dtype=np.float32)
    zero = lambda k: np.zeros((seg_len, 2), np.float32)       # noqa: E731
    iq = _spec_segments(n_segments, device, first_segment, seg_len, list(bits), [seg_len - 10485 * 100] * n_segments, "PSK", par, 2, zero)
    for j in range(n_segments):
        w = 0.1 * np.random.default_rng(42 + first_segment + j).normal(loc=0, scale=np.sqrt(2) / 2, size=(seg_len, 2))
        seg = iq[j * seg_len:(j + 1) * seg_len]
        seg.copy_((seg.to(torch.float64) + torch.from_numpy(w).to(seg.device)).to(torch.float32))
    return iq, bits
