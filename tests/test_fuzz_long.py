"""Long-form differential fuzzes of the state-carrying and order-sensitive kernels against the oracle (URH_FUZZ_ROUNDS rounds each,
default a handful inside `-m gpu`; URH_FUZZ_SEED picks another stream of cases): the Costas loop's speculative chunk evaluation
(signal_functions.pyx:252-330), the batched center statistics / plateau lengths (AutoInterpretation.py:226-277, :179-224) and the FIR
with its checked fallback (signal_functions.pyx:513-525)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROUNDS = int(os.environ.get("URH_FUZZ_ROUNDS", "4"))
SEED0 = int(os.environ.get("URH_FUZZ_SEED", "0"))


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def pipe():
    from urh_amd.pipeline import DevicePipeline
    return DevicePipeline()


def test_costas_fuzz(oracle):
    """random orders, sample types, carrier offsets, loop bandwidths, SNRs, gated pauses and un-gated noise stretches, lengths around the
    4096-sample chunk grid: the demodulated signal equals the serial recurrence from sample 1 on"""
    from urh_amd import signal_functions as sf
    for it in range(ROUNDS):
        rng = np.random.default_rng([313, SEED0, it])
        order = int(rng.choice([2, 4]))
        dtype = [np.float32, np.int16, np.int8, np.uint8, np.uint16][int(rng.integers(0, 5))]
        n = int(rng.choice([int(rng.integers(2, 9000)), 4096 * int(rng.integers(1, 40)) + int(rng.integers(-2, 3)), int(rng.integers(9000, 700_000))]))
        n = max(n, 2)
        sps = int(rng.choice([8, 64, 100, 500]))
        offset = float(rng.choice([0.0, 0.04, -0.006, 0.013, 0.2, float(rng.uniform(-0.3, 0.3))]))
        bw = float(rng.choice([0.1, 0.05, 0.25, 0.01, 0.5]))
        sym = rng.integers(0, order, n // sps + 1)
        phases = (np.array([-135, -45, 45, 135]) if order == 4 else np.array([-90, 90]))[sym] * np.pi / 180
        ph = np.repeat(phases, sps)[:n] + 2 * np.pi * offset * np.arange(n)
        iq = np.stack([np.cos(ph), np.sin(ph)], 1) + float(rng.choice([0.0, 0.07, 0.3, 1.0])) * np.sqrt(0.5) * rng.standard_normal((n, 2))
        for _ in range(int(rng.integers(0, 4))):                       # gated pauses / un-gated noise
            a, ln = int(rng.integers(0, n)), int(rng.choice([1, 50, 4096, 30_000]))
            if rng.random() < 0.5:
                iq[a:a + ln] *= 0.01
            else:
                iq[a:a + ln] = 0.5 * rng.standard_normal((len(iq[a:a + ln]), 2))
        if dtype == np.float32:
            iq, noise = iq.astype(np.float32), float(rng.choice([0.2, 0.0, 0.05]))
        else:
            info = np.iinfo(dtype)
            scale, off = (info.max - info.min) / 2 * 0.7, (info.max + info.min + 1) / 2
            iq = np.clip(np.round(iq * scale + off), info.min, info.max).astype(dtype)
            noise = 0.0 if np.dtype(dtype).kind == "u" else float(rng.choice([0.2, 0.0])) * scale
        want = oracle.afp_demod(iq, noise, "PSK", order, bw)
        got = sf.afp_demod(iq, noise, "PSK", order, bw)
        assert got.shape == want.shape and bits_equal(got[1:], want[1:]), \
            (it, SEED0, order, np.dtype(dtype).name, n, sps, offset, bw, noise, int((got[1:].view(np.uint32) != want[1:].view(np.uint32)).sum()))


def test_center_and_plateau_fuzz(pipe, oracle):
    """random rectangular signals (two or four levels, noise, gated samples, NaN / inf sprinkled rarely, constant stretches) cut into random
    messages: every message's center equals the oracle's detect_center as a float, its plateau lengths equal get_plateau_lengths"""
    import torch
    from urh_amd import estimators
    for it in range(ROUNDS):
        rng = np.random.default_rng([515, SEED0, it])
        n = int(rng.choice([int(rng.integers(10, 5000)), int(rng.integers(5000, 300_000)), int(rng.integers(300_000, 2_000_000))]))
        sps = int(rng.choice([7, 40, 50, 300]))
        levels = int(rng.choice([2, 4]))
        amp, off = float(rng.choice([1.1, 0.9, 3.0, 1e-3])), float(rng.choice([-0.55, 0.05, -1.5, 0.0]))
        qad = (np.repeat(rng.integers(0, levels, n // sps + 1), sps)[:n] * amp / (levels - 1) + off
               + float(rng.choice([0.0, 0.03, 0.08, 0.4])) * amp * rng.standard_normal(n)).astype(np.float32)
        if rng.random() < 0.7:
            qad[rng.integers(0, n, int(rng.integers(0, max(n // 50, 1))))] = -4.0
        for _ in range(int(rng.integers(0, 3))):
            a, ln = int(rng.integers(0, n)), int(rng.choice([1, 100, 5000, 70_000]))
            qad[a:a + ln] = float(rng.choice([-4.0, 0.25, -4.0]))
        if rng.random() < 0.1:
            qad[int(rng.integers(0, n))] = np.float32(rng.choice([np.nan, np.inf]))
        k = int(rng.integers(1, 14))
        cuts = np.sort(rng.integers(0, n + 1, 2 * k))
        bounds = [(int(cuts[2 * i]), int(cuts[2 * i + 1])) for i in range(k)]
        if rng.random() < 0.15:
            bounds = [(0, n)]                                          # the whole signal as one message (what detect_center itself sees)
        dev = torch.from_numpy(qad).cuda()
        try:
            centers = estimators.centers_batched(pipe, dev, bounds)
        except Exception as exc:
            raise AssertionError((it, SEED0, n, bounds, repr(exc))) from exc
        with np.errstate(all="ignore"):
            for (a, b), c in zip(bounds, centers):
                want = oracle.detect_center(qad[a:b]) if b > a else None
                same = (c is None and want is None) or (c is not None and want is not None and
                                                        (float(c) == float(want) or (np.isnan(c) and np.isnan(want))))
                assert same, (it, SEED0, n, (a, b), c, want)
        use = [c if (c is not None and np.isfinite(c)) else (float(rng.choice([0.1, 0.0, -0.3])) if i % 2 else None) for i, c in enumerate(centers)]
        plats = estimators.plateau_lengths_batched(pipe, dev, bounds, use)
        for (a, b), c, got in zip(bounds, use, plats):
            want = oracle.get_plateau_lengths(qad[a:b], c, 25) if c is not None else np.zeros(0, np.uint64)
            assert np.array_equal(np.asarray(got, dtype=np.uint64), np.asarray(want, dtype=np.uint64)), (it, SEED0, n, (a, b), c, len(got), len(want))
        # the chained call equals the two calls
        r = np.array(bounds, np.int64).reshape(-1, 2)
        c1, t1, b1 = estimators.centers_and_decisions(pipe, dev, r, 25)
        c2 = estimators.centers_array(pipe, dev, r)
        t2, b2 = estimators._plateau_decisions(pipe, dev, r, c2.astype(np.float32).astype(np.float64), 25)
        assert np.array_equal(c1, c2, equal_nan=True) and np.array_equal(t1, t2) and np.array_equal(b1, b2), (it, SEED0, n, bounds)


def test_fir_fuzz(oracle):
    """random lengths, tap counts (1 .. 300), magnitudes up to the checked kernel's hand-over threshold and beyond, inf / NaN sprinkled"""
    from urh_amd import signal_functions as sf
    for it in range(ROUNDS):
        rng = np.random.default_rng([717, SEED0, it])
        n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20_000)), int(rng.integers(20_000, 600_000))]))
        m = int(rng.choice([1, 2, 3, 10, 63, 64, 65, int(rng.integers(1, 300))]))
        mag = float(rng.choice([1.0, 1e-20, 1e18, 1e30]))
        x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * mag).astype(np.complex64)
        h = ((rng.standard_normal(m) + 1j * rng.standard_normal(m)) * float(rng.choice([1.0, 0.05, 1e10]))).astype(np.complex64)
        for _ in range(int(rng.integers(0, 3))):
            x[int(rng.integers(0, n))] = np.complex64(rng.choice([np.inf, -np.inf, np.nan, 0.0]) + 1j * rng.choice([0.0, np.inf, np.nan, 1.0]))
        with np.errstate(all="ignore"):
            want = oracle.fir_filter(x, h)
        got = sf.fir_filter(x, h)
        a, b = np.ascontiguousarray(got, np.complex64).view(np.uint32), np.ascontiguousarray(want, np.complex64).view(np.uint32)
        fa, fb = a.view(np.float32), b.view(np.float32)
        assert bool(((a == b) | (np.isnan(fa) & np.isnan(fb))).all()), (it, SEED0, n, m, mag, int(((a != b) & ~(np.isnan(fa) & np.isnan(fb))).sum()))


def test_estimate_fuzz(pipe):
    """AutoInterpretation.estimate (AutoInterpretation.py:373-470) end to end against the REAL reference (oracle/ref_python.py) on random
    bursty captures: FSK and OOK / ASK, random message lengths, pauses, symbol lengths, SNRs, sample types; modulation given or detected,
    noise given or automatic.  The dict -- modulation, bit length, tolerance, center and noise as floats -- must be identical."""
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_python
    if not ref_python.available():
        pytest.skip("oracle/_ref (compiled reference + Python sources) not present")
    ref_python.setup()
    from urh.ainterpretation import AutoInterpretation as AI
    from urh_amd import estimators
    for it in range(ROUNDS):
        rng = np.random.default_rng([919, SEED0, it])
        mod = str(rng.choice(["FSK", "OOK", "ASK"]))
        sps = int(rng.choice([8, 20, 50, 100, 250]))
        n_msgs = int(rng.integers(1, 9))
        sigma = float(rng.choice([0.005, 0.02, 0.06]))
        parts = [sigma * rng.standard_normal((int(rng.integers(200, 4000)), 2))]
        for _ in range(n_msgs):
            nsym = int(rng.integers(12, 200))
            bits = rng.integers(0, 2, nsym)
            bits[:8] = [1, 0, 1, 0, 1, 0, 1, 0]
            t = np.arange(nsym * sps)
            if mod == "FSK":
                f = np.repeat(np.where(bits == 1, 0.03, -0.03), sps)
                ph = 2 * np.pi * np.cumsum(f)
                msg = np.stack([np.cos(ph), np.sin(ph)], 1)
            else:
                env = np.repeat(np.where(bits == 1, 1.0, 0.0 if mod == "OOK" else 0.3), sps)
                ph = 2 * np.pi * 0.011 * t
                msg = np.stack([env * np.cos(ph), env * np.sin(ph)], 1)
            amp = float(rng.choice([1.0, 0.6, 0.35]))
            parts.append(amp * msg + sigma * rng.standard_normal(msg.shape))
            parts.append(sigma * rng.standard_normal((int(rng.choice([sps * 15, sps * 40, 3000, 20_000])), 2)))
        x = np.concatenate(parts)
        dtype = [np.float32, np.float32, np.int8, np.int16][int(rng.integers(0, 4))]
        if dtype == np.float32:
            iq = x.astype(np.float32)
        else:
            info = np.iinfo(dtype)
            iq = np.clip(np.round(x * info.max * 0.7), info.min, info.max).astype(dtype)
        given_mod = None if rng.random() < 0.4 else mod
        with np.errstate(all="ignore"):
            want = AI.estimate(iq, noise=None, modulation=given_mod)
        if given_mod is None:
            dev_iq = torch.from_numpy(iq).cuda()
            nz = estimators.detect_noise_level_dev(pipe, dev_iq)
            if estimators.detect_modulation_for_messages_dev(dev_iq, estimators.segment_messages_dev(pipe, dev_iq, nz)) == "PSK":
                continue                 # the reference's Costas demodulator leaves result[0] uninitialised: not reproducible (see the golden test)
        got = estimators.estimate_dev(pipe, torch.from_numpy(iq).cuda(), noise=None, modulation=given_mod)
        tag = (it, SEED0, mod, given_mod, sps, n_msgs, sigma, np.dtype(dtype).name, len(iq))
        if want is None:
            assert got is None, (tag, got)
            continue
        assert got is not None, (tag, want)
        assert got["modulation_type"] == want["modulation_type"] and int(got["bit_length"]) == int(want["bit_length"]) \
            and int(got["tolerance"]) == int(want["tolerance"]), (tag, got, want)
        assert float(got["center"]) == float(want["center"]) and float(got["noise"]) == float(want["noise"]), (tag, got, want)


def test_fsk_loops_fuzz(pipe, oracle):
    """the hot kernel's fast loop, wide loop and per-row forms handing over to one another (signal_functions.pyx:363-376): stretches of
    random phase steps from a hundredth of a radian to beyond pi, random amplitudes (|re| at the division's window), noise, gated
    pauses, exact zeros and spikes, lengths around the 64-row chunk grid -- the demodulated signal bit for bit, the pulse table equal"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    pipe_wide = DevicePipeline(0, tuning={"wide_int": 1})      # signed integer captures through the instantiation with the wide loop as well
    from urh_amd import _lib as _l
    wide0, wide_want = _l.load().urhgpu_test_wide_int_launches(), 0
    for it in range(4 * ROUNDS):
        rng = np.random.default_rng([717, SEED0, it])
        n = int(rng.choice([8192 * int(rng.integers(1, 60)) + int(rng.choice([0, 1, 127, 128, 2049])), int(rng.integers(2000, 900_000))]))
        steps = []
        while sum(len(s) for s in steps) < n:
            k = int(rng.choice([3, 40, 300, 3000, 20_000]))
            s = float(rng.choice([0.01, 0.1, 0.3, 0.41, 0.45, 0.7, 1.0, 1.5, 1.6, 2.4, 3.0, 3.14159, float(rng.uniform(0, 3.2))]))
            steps.append(np.repeat(rng.choice([-s, s], k // int(rng.choice([1, 4, 40])) + 1), int(rng.choice([1, 4, 40])))[:k])
        ph = np.cumsum(np.concatenate(steps)[:n])
        amp = np.ones(n)
        for _ in range(int(rng.integers(0, 6))):
            a, ln = int(rng.integers(0, n)), int(rng.choice([1, 2, 64, 257, 5000, 40_000]))
            amp[a:a + ln] = float(rng.choice([0.0, 0.01, 1e-15, 1e12, 3.0]))
        x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1) + float(rng.choice([0.0, 0.01, 0.05, 0.3])) * rng.standard_normal((n, 2)) * (amp[:, None] > 0)
        dtype = [np.float32, np.float32, np.int16, np.int8][it % 4]
        if dtype == np.float32:
            iq, scale = x.astype(np.float32), 1.0
        else:
            scale = float(rng.choice([0.6, 0.05])) * np.iinfo(dtype).max
            iq = np.clip(np.round(np.clip(x, -1.5, 1.5) * scale), np.iinfo(dtype).min, np.iinfo(dtype).max).astype(dtype)
        noise = float(rng.choice([0.0, 0.0, 0.1, 0.5])) * scale
        tol, sps = int(rng.choice([0, 1, 3, 5])), int(rng.choice([1, 4, 40]))
        p = DemodParams("FSK", 1, noise, float(rng.choice([0.0, 0.3, -1.0])), 1.0, tol, sps, 0.1, 8, True)
        qad = oracle.afp_demod(iq, noise, "FSK", 2)
        pp = oracle.grab_pulse_lens(qad, p.center, tol, "FSK", sps, 1, 1.0)
        from urh_amd import _lib
        lib = _lib.load()
        for tiles, pl in [(0, pipe), (4, pipe)] + ([(0, pipe_wide), (4, pipe_wide)] if dtype != np.float32 else []):
            # the default chunk plan (short chunks for short captures) and the benchmark's 64-row chunks
            assert lib.urhgpu_test_force_tiles_per_chunk(tiles) == 0
            try:
                res = pl.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // (tol + 1) + 2)
                got = res.qad.cpu().numpy()
                rows = res.ppseq()
            finally:
                lib.urhgpu_test_force_tiles_per_chunk(0)
            bad = np.nonzero(got.view(np.uint32) != qad.view(np.uint32))[0]
            assert len(bad) == 0, (it, SEED0, tiles, pl is pipe_wide, n, np.dtype(dtype).name, noise, len(bad), bad[:6], got[bad[:6]], qad[bad[:6]])
            assert np.array_equal(rows, pp), (it, SEED0, tiles, pl is pipe_wide, n, np.dtype(dtype).name)
            wide_want += int(pl is pipe_wide and n >= 2048)
    launched = _l.load().urhgpu_test_wide_int_launches() - wide0
    assert wide_want == 0 or launched >= wide_want // 2, (launched, wide_want)       # the keyed passes did take the wide instantiation
