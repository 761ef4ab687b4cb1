"""modulate_c (signal_functions.pyx:56-177): the oracle against the real reference's outputs (tests/golden/modulate, made
by tests/golden/make_modulate_golden.py) and -- where oracle/_ref is built -- against the reference itself on random
arguments; the HIP generator against both (GPU tests), bit-exact for every sample type."""
import array
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "modulate", "modulate.npz")


def golden_cases():
    g = np.load(GOLD)
    for entry in g["names"]:
        name, mod, dtype = str(entry).split(":")
        bps, sps, pause, start = (int(v) for v in g[name + "_args"])
        amp, freq, phase, rate = (float(v) for v in g[name + "_amp"])
        yield name, (g[name + "_bits"], sps, mod, g[name + "_par"], bps, amp, freq, phase, rate, pause, start, np.dtype(dtype).type), g[name + "_out"]


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def random_cases(seed, n):
    rng = np.random.default_rng(seed)
    mods = [("FSK", [-20e3, 20e3], 1), ("FSK", [-30e3, -10e3, 10e3, 30e3], 2), ("ASK", [0.0, 1.0], 1), ("ASK", [0.0, 0.3, 0.6, 1.0], 2),
            ("PSK", [-np.pi / 2, np.pi / 2], 1), ("PSK", list(np.array([-135.0, -45.0, 45.0, 135.0]) * np.pi / 180), 2),
            ("FSK", list(np.linspace(-35e3, 35e3, 8)), 3), ("OQPSK", list(np.array([-135.0, -45.0, 45.0, 135.0]) * np.pi / 180), 2)]
    for k in range(n):
        mod, par, bps = mods[k % len(mods)]
        dtype = (np.float32, np.int8, np.int16)[(k // len(mods)) % 3]
        amp = {np.float32: 1.0, np.int8: 127.0, np.int16: 32767.0}[dtype]
        if mod == "ASK":
            par = [p * amp for p in par]
        sps = int(rng.choice([1, 7, 8, 100, 333]))
        nb = int(rng.integers(0, 700)) * bps + int(rng.integers(0, bps))        # trailing bits that fill no symbol are ignored
        pause = int(rng.choice([0, 1, 76, 5000]))
        if mod == "OQPSK" and nb < 2 and pause < sps:
            pause = sps                                         # the reference's blanking loops index past the array otherwise
        start = int(rng.choice([0, 17, 123_456, 16_777_217, 3_000_000_000]))
        bits = rng.integers(0, 2, nb).astype(np.uint8)
        yield (bits, sps, mod, np.array(par, np.float32), bps, amp, 40e3, float(rng.uniform(-3, 3)), float(rng.choice([1e6, 2e6, 250e3])),
               pause, start, dtype)


def test_oracle_equals_reference_goldens(oracle):
    n = 0
    for name, args, want in golden_cases():
        assert same(oracle.modulate_c(*args), want), name
        n += 1
    assert n == 9


def test_oracle_equals_real_reference_random(oracle):
    import build_ref
    if not build_ref.built():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    sf, _, _ = build_ref.import_ref()
    for args in random_cases(5, 96):
        bits, sps, mod, par = args[:4]
        ref = sf.modulate_c(array.array("B", bits.tolist()), sps, mod, array.array("f", par.tolist()), *args[4:])
        assert same(oracle.modulate_c(*args), ref), args[1:]


# ---- GFSK (signal_functions.pyx:196-243) ---------------------------------------------------------------------------
def gfsk_golden_cases():
    g = np.load(GOLD)
    for entry in g["gfsk_names"]:
        name, dtype = str(entry).split(":")
        bps, sps, pause, start = (int(v) for v in g[name + "_args"])
        amp, phase, rate, bt, fw = (float(v) for v in g[name + "_amp"])
        yield name, dict(bits=g[name + "_bits"], sps=sps, par=g[name + "_par"], bps=bps, amp=amp, phase=phase, rate=rate, pause=pause,
                         start=start, dtype=np.dtype(dtype).type, bt=bt, fw=fw), g[name + "_freqs"], g[name + "_gfir"], g[name + "_out"]


def gfsk_random_cases(seed, n):
    rng = np.random.default_rng(seed)
    for k in range(n):
        bps = 1 + k % 2
        par = np.array([-20e3, 20e3] if bps == 1 else [-30e3, -10e3, 10e3, 30e3], np.float32)
        dtype = (np.float32, np.int8, np.int16)[k % 3]
        sps = int(rng.choice([4, 8, 25, 100]))
        yield dict(bits=rng.integers(0, 2, int(rng.integers(1, 120)) * bps).astype(np.uint8), sps=sps, par=par, bps=bps,
                   amp={np.float32: 1.0, np.int8: 127.0, np.int16: 32767.0}[dtype], phase=float(rng.uniform(-3, 3)),
                   rate=float(rng.choice([1e6, 2e6])), pause=int(rng.choice([0, 76, 1000])),
                   start=int(rng.choice([0, 17, 123_456, 16_777_217, 3_000_000_000])), dtype=dtype,
                   bt=float(rng.choice([0.3, 0.5, 1.0])), fw=float(rng.choice([1.0, 2.0, 5.0])))


def oracle_gfsk(oracle, c, **kw):
    return oracle.modulate_gfsk(c["bits"], c["sps"], c["par"], c["bps"], c["amp"], c["phase"], c["rate"], c["pause"], c["start"],
                                c["dtype"], c["bt"], c["fw"], **kw)


def test_oracle_gfsk_equals_reference_goldens(oracle):
    """Everything downstream of the Gaussian convolution is bit-exact against the real reference when it is fed numpy's
    frequencies (recorded with the vectors); the restated convolution (exact dot products) is within 2 float32 ulps of the
    largest symbol frequency of numpy's BLAS one."""
    n = 0
    for name, c, freqs, gfir, want in gfsk_golden_cases():
        assert np.array_equal(oracle.gauss_fir(c["rate"], c["sps"], c["bt"], c["fw"]), gfir), name
        assert same(oracle_gfsk(oracle, c, frequencies=freqs), want), name
        _, fr, ph = oracle_gfsk(oracle, c, return_freqs_phases=True)
        assert np.abs(fr - freqs).max() <= 2 * np.spacing(np.abs(c["par"]).max()), name
        n += 1
    assert n == 4


def test_oracle_gfsk_equals_real_reference_random(oracle):
    import build_ref
    if not build_ref.built():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from urh_amd.signal_functions import gfsk_frequencies_numpy
    sf, _, _ = build_ref.import_ref()
    for c in gfsk_random_cases(8, 40):
        ref = sf.modulate_c(array.array("B", c["bits"].tolist()), c["sps"], "GFSK", array.array("f", c["par"].tolist()), c["bps"], c["amp"],
                            40e3, c["phase"], c["rate"], c["pause"], c["start"], c["dtype"], c["bt"], c["fw"])
        freqs = gfsk_frequencies_numpy(c["bits"], c["par"], c["sps"], c["bps"], oracle.gauss_fir(c["rate"], c["sps"], c["bt"], c["fw"]))
        assert same(oracle_gfsk(oracle, c, frequencies=freqs), ref), {k: v for k, v in c.items() if k != "bits"}


def gpu_gfsk(sf, c, **kw):
    return sf.modulate_c(c["bits"], c["sps"], "GFSK", c["par"], c["bps"], c["amp"], 40e3, c["phase"], c["rate"], c["pause"], c["start"],
                         c["dtype"], c["bt"], c["fw"], **kw)


@pytest.mark.gpu
def test_gpu_gfsk_equals_goldens_and_oracle(oracle):
    """Fed the recorded numpy frequencies the GPU generator reproduces the real reference bit for bit; with its own convolution
    it equals the oracle's restatement bit for bit (filters longer than the message, late starts, every sample type)."""
    from urh_amd import signal_functions as sf
    for name, c, freqs, gfir, want in gfsk_golden_cases():
        assert np.array_equal(sf.gauss_fir(c["rate"], c["sps"], c["bt"], c["fw"]), gfir), name
        assert same(gpu_gfsk(sf, c, gfsk_frequencies=freqs), want), name
        assert same(gpu_gfsk(sf, c), oracle_gfsk(oracle, c)), name
    for c in gfsk_random_cases(9, 60):
        assert same(gpu_gfsk(sf, c), oracle_gfsk(oracle, c)), {k: v for k, v in c.items() if k != "bits"}
    c = next(gfsk_random_cases(10, 1))
    assert same(gpu_gfsk(sf, c, gfsk_frequencies="numpy"),
                oracle_gfsk(oracle, c, frequencies=sf.gfsk_frequencies_numpy(c["bits"], c["par"], c["sps"], c["bps"],
                                                                             sf.gauss_fir(c["rate"], c["sps"], c["bt"], c["fw"]))))


@pytest.mark.gpu
def test_gpu_gfsk_batch_and_roundtrip(oracle):
    """Batched GFSK equals the single calls; /root/reference/tests/test_modulator.py:68-86 (test_gfsk): three GFSK messages,
    demodulated as FSK, give their bits back."""
    from urh_amd import signal_functions as sf
    rng = np.random.default_rng(12)
    msgs = [rng.integers(0, 2, n).astype(np.uint8) for n in (300, 1, 2000, 64)]
    pauses = [76, 10, 300, 5000]
    par = np.array([-10e3, 20e3], np.float32)
    got = sf.modulate_messages_dev(msgs, 100, "GFSK", par, 1, 1.0, 40e3, 0.1, 1e6, pauses, None, np.float32).cpu().numpy()
    off = 0
    for m, p in zip(msgs, pauses):
        want = oracle.modulate_gfsk(m, 100, par, 1, 1.0, 0.1, 1e6, p, off, np.float32)
        assert same(got[off:off + len(want)], want)
        off += len(want)
    assert off == len(got)
    iq = np.concatenate([sf.modulate_c(b, 100, "GFSK", par, 1, 1.0, 40e3, 0.0, 1e6, p, 0)
                         for b, p in (([1, 0, 0, 1, 0], 9437), ([1, 0, 1], 9845), ([1, 0, 1, 0], 8458))])
    qad = sf.afp_demod(iq, 0.02, "FSK", 2)
    pp = sf.grab_pulse_lens(qad, 0.03, 5, "FSK", 100)
    data, _, _ = sf.ppseq_to_bits(pp, 100, 1)
    assert ["".join(map(str, d)) for d in data] == ["10010", "101", "1010"]


@pytest.mark.gpu
def test_gpu_modulate_equals_goldens_and_oracle(oracle):
    from urh_amd import signal_functions as sf
    for name, args, want in golden_cases():
        assert same(sf.modulate_c(*args), want), name
    for args in random_cases(11, 144):
        assert same(sf.modulate_c(*args), oracle.modulate_c(*args)), args[1:]


@pytest.mark.gpu
def test_gpu_modulate_long_message_and_batch(oracle):
    """2^20-sample messages (carrier arguments up to 1.3e5 rad: glibc's large-argument reduction) and the batched entry
    point: messages rendered back to back by one launch equal the single calls."""
    from urh_amd import signal_functions as sf
    rng = np.random.default_rng(3)
    msgs = [rng.integers(0, 2, n).astype(np.uint8) for n in (10485, 0, 1, 2000, 5000)]
    pauses = [76, 10, 0, 300, 48576]
    for mod, par, dtype in (("FSK", [-20e3, 20e3], np.float32), ("ASK", [0.0, 100.0], np.int8), ("PSK", [-1.0, 2.0], np.int16),
                            ("OQPSK", [-2.0, -1.0, 1.0, 2.0], np.float32)):
        amp = {np.float32: 1.0, np.int8: 127.0, np.int16: 32767.0}[dtype]
        bps = 2 if mod == "OQPSK" else 1
        for starts in (None, [0, 5, 5, 1 << 24, 77]):
            got = sf.modulate_messages_dev(msgs, 100, mod, par, bps, amp, 40e3, 0.1, 1e6, pauses, starts, dtype).cpu().numpy()
            off = 0
            for k, (m, p) in enumerate(zip(msgs, pauses)):
                st = off if starts is None else starts[k]
                want = oracle.modulate_c(m, 100, mod, np.array(par, np.float32), bps, amp, 40e3, 0.1, 1e6, p, st, dtype)
                assert same(got[off:off + len(want)], want), (mod, dtype, k)
                off += len(want)
            assert off == len(got)


@pytest.mark.gpu
def test_gpu_modulate_then_demodulate_roundtrip():
    """/root/reference/tests/test_demodulations.py:55-72: modulate '101010' FSK at 8 samples/symbol, demodulate, read the bits back."""
    from urh_amd import signal_functions as sf
    bits = array.array("B", [1, 0, 1, 0, 1, 0] * 20)
    iq = sf.modulate_c(bits, 8, "FSK", array.array("f", [-10e3, 10e3]), 1, 1.0, 40e3, 0.0, 1e6, 0, 0)
    qad = sf.afp_demod(iq, 0.0, "FSK", 2)
    assert qad.max() < 1
    pp = sf.grab_pulse_lens(qad, 0.0, 0, "FSK", 8)
    data, _, _ = sf.ppseq_to_bits(pp, 8, 1)
    got = "".join(map(str, data[0]))
    assert got.startswith("101010") and len(got) >= 118


@pytest.mark.gpu
def test_gpu_modulate_errors_and_edges():
    from urh_amd import signal_functions as sf
    assert sf.modulate_c(array.array("B"), 100, "FSK", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 50, 0).shape == (50, 2)
    assert not sf.modulate_c(array.array("B"), 100, "FSK", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 50, 0).any()
    with pytest.raises(ValueError):
        sf.modulate_c([1, 0], 10, "FSK", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 0, 0, dtype=np.uint8)
    with pytest.raises(ZeroDivisionError):                                                       # len(bits) // num_symbols (:201)
        sf.modulate_c([1], 10, "GFSK", [-1.0, 1.0, 2.0, 3.0], 2, 1.0, 0.0, 0.0, 1e6, 0, 0)
    assert sf.modulate_c([], 10, "GFSK", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 7, 0).shape == (7, 2)
    with pytest.raises(AssertionError):
        sf.modulate_c([1, 0], 10, "OQPSK", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 0, 0)          # bits_per_symbol must be 2
    with pytest.raises(AssertionError):
        sf.modulate_c([1, 0], 10, "QAM", [-1.0, 1.0], 1, 1.0, 0.0, 0.0, 1e6, 0, 0)
