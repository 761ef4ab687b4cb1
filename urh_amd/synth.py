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
