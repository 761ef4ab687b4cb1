"""Parity tests proper: the HIP path (through the C ABI) against the committed golden vectors of the
real reference and against the oracle on seeded inputs.  Bit-exact everywhere: integer tables, bit
strings AND the float32 demodulated signal (ASK: IEEE ops only; FSK: fdlibm atan2f restated on the GPU).
Run on the GPU box:  python -m pytest tests -m gpu
"""
import ctypes as C

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, load_golden, synth_fsk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf():
    from urh_amd import signal_functions
    return signal_functions


@pytest.fixture(scope="module")
def pipe():
    from urh_amd.pipeline import DevicePipeline
    return DevicePipeline()


def bits_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_device_atan2f_is_glibc_atan2f(oracle):
    import torch
    from urh_amd import _lib
    rng = np.random.default_rng(0)
    n = 1 << 22
    raw = rng.integers(0, 2 ** 32, size=(4, n), dtype=np.uint64).astype(np.uint32)
    y = np.concatenate([raw[0].view(np.float32), (rng.standard_normal(n) * 0.3).astype(np.float32),
                        np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 3e38], np.float32)])
    x = np.concatenate([raw[1].view(np.float32), (1 + rng.standard_normal(n) * 0.3).astype(np.float32),
                        np.array([0.0, 0.0, -0.0, -0.0, 0.0, 0.0, np.inf, -np.inf, 1.0, 3e38, 1e-45], np.float32)])
    want = oracle.atan2f(y, x)
    dy, dx = torch.from_numpy(y).cuda(), torch.from_numpy(x).cuda()
    out = torch.empty_like(dy)
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().urhgpu_test_atan2f_dev(ctx.handle, C.c_void_p(dy.data_ptr()), C.c_void_p(dx.data_ptr()),
                                                  len(y), C.c_void_p(out.data_ptr())))
    got = out.cpu().numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{(~same).sum()} of {len(y)} differ"


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_host_api(sf, name):
    g = load_golden(name)
    mod, bps = g["modulation_type"], g["bits_per_symbol"]
    qad = sf.afp_demod(g["iq"], g["noise_threshold"], mod, 2 ** bps, g["costas_loop_bandwidth"])
    st = 1 if mod == "PSK" else 0             # the reference leaves result[0] of the Costas loop uninitialised
    assert bits_equal(qad[st:], g["qad"][st:]), int((qad[st:].view(np.uint32) != g["qad"][st:].view(np.uint32)).sum())
    pp = sf.grab_pulse_lens(g["qad"], g["center"], g["tolerance"], mod, g["samples_per_symbol"], bps, g["center_spacing"])
    assert np.array_equal(pp, g["ppseq"])
    bits, off, pauses, pos, poff = sf.ppseq_to_bits_flat(g["ppseq"], g["samples_per_symbol"], bps, True, g["pause_threshold"])
    assert np.array_equal(bits, g["bits"]) and np.array_equal(off, g["msg_off"])
    assert np.array_equal(pauses, g["pauses"]) and np.array_equal(pos, g["pos"]) and np.array_equal(poff, g["pos_off"])
    if g["kat"]:
        first = "".join(map(str, bits[off[0]:off[1]]))
        assert first == g["kat"] if g["kat_mode"] == "exact" else first.startswith(g["kat"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_fused_device_path(pipe, name):
    import torch
    from urh_amd.pipeline import DemodParams
    g = load_golden(name)
    p = DemodParams(g["modulation_type"], g["bits_per_symbol"], g["noise_threshold"], g["center"], g["center_spacing"],
                    g["tolerance"], g["samples_per_symbol"], g["costas_loop_bandwidth"], g["pause_threshold"], True)
    iq = torch.from_numpy(g["iq"]).cuda()
    res = pipe.iq_to_bits(iq, p, want_qad=True)
    if g["modulation_type"] == "PSK":
        # the golden qad's element 0 is whatever np.empty held in the reference run; the table is compared against
        # the oracle's (which segments the same signal with element 0 = -4, like the GPU path)
        assert bits_equal(res.qad.cpu().numpy()[1:], g["qad"][1:])
        return
    assert bits_equal(res.qad.cpu().numpy(), g["qad"])
    assert np.array_equal(res.ppseq(), g["ppseq"])
    bits, off, pauses, pos, poff = res.flat()
    assert np.array_equal(bits, g["bits"]) and np.array_equal(off, g["msg_off"])
    assert np.array_equal(pauses, g["pauses"]) and np.array_equal(pos, g["pos"]) and np.array_equal(poff, g["pos_off"])
    # bits-only mode (qad not materialised) gives the same table
    res2 = pipe.iq_to_bits(iq, p, want_qad=False)
    assert np.array_equal(res2.ppseq(), g["ppseq"])


SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 511, 512, 513, 1023, 8191, 8192, 8193, 16384 + 7, 70001, 300000]


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
def test_afp_demod_sizes_and_dtypes(sf, oracle, mod):
    for dtype in (np.float32, np.int8, np.uint8, np.int16, np.uint16):
        for n in SIZES:
            iq = synth_fsk(n, sps=20, seed=n, noise=0.1, pause_every=500, pause_len=90, dtype=dtype)
            scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64)).max())
            for noise in (0.0, 0.4 * scale):
                want = oracle.afp_demod(iq, noise, mod, 2)
                got = sf.afp_demod(iq, noise, mod, 2)
                assert bits_equal(got, want), (mod, np.dtype(dtype).name, n, noise, int((got != want).sum()))


@pytest.mark.parametrize("mod", ["FSK", "ASK", "QAM"])
def test_afp_demod_streaming_kernel_and_remainder(sf, oracle, mod):
    """urhgpu_afp_demod sends whole 8192-sample chunks through the hot kernel's streaming structure (no run phase) and the
    remainder through k_afp_demod: sizes either side of the chunk size, every sample type."""
    for dtype in (np.float32, np.int8, np.uint16):
        for n in (8191, 8192, 8193, 8192 * 3 + 1, 8192 * 5 + 4099, 100_000):
            iq = synth_fsk(n, sps=25, seed=n + 1, noise=0.1, pause_every=3000, pause_len=400, dtype=dtype)
            scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64)).max())
            for noise in (0.0, 0.4 * scale):
                want = oracle.afp_demod(iq, noise, mod, 2)
                got = sf.afp_demod(iq, noise, mod, 2)
                assert bits_equal(got, want), (mod, np.dtype(dtype).name, n, noise, int((got != want).sum()))


def test_afp_demod_signed_zero_and_nonfinite(sf, oracle):
    vals = np.array([0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40, 1e-30, 3e38, -3e38, 0.5, -0.25], dtype=np.float32)
    rng = np.random.default_rng(5)
    iq = vals[rng.integers(0, len(vals), size=(100000, 2))]
    for mod in ("FSK", "ASK"):
        want = oracle.afp_demod(iq, 0.0, mod, 2)
        got = sf.afp_demod(iq, 0.0, mod, 2)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (mod, int((~same).sum()))


def _rect(rng, n, mod, bps, noise_val):
    levels = rng.choice([-1.0, -0.3, 0.3, 1.0] if bps == 2 else [-0.5, 0.5], size=n // max(1, int(rng.integers(1, 60))) + 1)
    x = np.repeat(levels, n // len(levels) + 1)[:n].astype(np.float32)
    x[rng.random(n) < rng.choice([0, 0.02, 0.1, 0.3])] *= -1
    for _ in range(int(rng.integers(0, 5))):
        a = int(rng.integers(0, n))
        x[a:min(n, a + int(rng.integers(1, 3000)))] = noise_val
    if mod == "ASK":
        x = np.abs(x)
    return x


def test_grab_pulse_lens_randomised(sf, oracle):
    rng = np.random.default_rng(3)
    for it in range(150):
        n = int(rng.choice(SIZES + [40000, 123457]))
        mod = ["ASK", "FSK", "PSK"][it % 3]
        bps = int(rng.integers(1, 4))
        tol = int(rng.choice([0, 1, 2, 5, 7, 33, 100, 3000, 20000]))
        sps = int(rng.choice([3, 8, 100]))
        x = _rect(rng, n, mod, min(bps, 2), oracle.noise_for_mod_type(mod))
        center, spacing = (0.0 if mod != "ASK" else 0.4), float(rng.choice([0.1, 0.6]))
        want = oracle.grab_pulse_lens(x, center, tol, mod, sps, bps, spacing)
        got = sf.grab_pulse_lens(x, center, tol, mod, sps, bps, spacing)
        assert np.array_equal(want, got), (it, n, mod, bps, tol, sps, len(want), len(got))
        for pt in (0, 8):
            fb = oracle.ppseq_to_bits_flat(want, sps, bps, True, pt)
            gb = sf.ppseq_to_bits_flat(got, sps, bps, True, pt)
            assert all(np.array_equal(a, b) for a, b in zip(fb, gb)), (it, pt)


def test_empty_and_tiny_inputs(sf):
    assert sf.afp_demod(np.zeros((0, 2), np.float32), 0.0, "FSK", 2).shape == (0,)
    assert sf.afp_demod(np.ones((2, 2), np.float32), 0.0, "FSK", 2).tolist() == [0.0, 0.0]       # n <= 2 -> zeros
    assert sf.grab_pulse_lens(np.zeros(0, np.float32), 0.0, 5, "FSK", 100).shape == (0, 2)
    with pytest.raises(ValueError):
        sf.afp_demod(np.zeros((10, 2), np.float64), 0.0, "FSK", 2)                                # Unsupported dtype
    b = sf.ppseq_to_bits(np.zeros((0, 2), np.int64), 100, 1)
    assert b[0] == [] and list(b[1]) == [] and b[2] == []


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
def test_fused_equals_oracle_medium(pipe, oracle, mod):
    """1 M-sample noisy capture with silent gaps: fused device path vs the oracle, everything bit-exact."""
    import torch
    from urh_amd.pipeline import DemodParams
    n = 1 << 20
    iq = synth_fsk(n, sps=100, seed=21, noise=0.05, pause_every=150000, pause_len=9000)
    if mod == "ASK":
        rng = np.random.default_rng(4)
        env = np.repeat(rng.integers(0, 2, n // 100 + 1), 100)[:n].astype(np.float32)
        iq = (iq * (0.05 + 0.95 * env)[:, None]).astype(np.float32)
    for noise, tol, center in ((0.0, 5, 0.0 if mod == "FSK" else 0.35), (0.2, 2, 0.0 if mod == "FSK" else 0.35)):
        p = DemodParams(mod, 1, noise, center, 1.0, tol, 100, 0.1, 8, True)
        qad = oracle.afp_demod(iq, noise, mod, 2)
        pp = oracle.grab_pulse_lens(qad, center, tol, mod, 100, 1, 1.0)
        fb = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
        res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True)
        assert bits_equal(res.qad.cpu().numpy(), qad)
        assert np.array_equal(res.ppseq(), pp)
        assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))


def test_fused_randomised_vs_oracle(pipe, oracle, cases=None, big=False):
    """Differential fuzz of the fused device path against the oracle: random sizes (chunk and row boundaries fall anywhere in
    runs and pauses), tolerances, samples per symbol, noise gates, centers, sample types; everything bit-exact.
    big: sizes up to 600 000 samples (dozens of 64-row chunks under a forced chunk plan)."""
    import torch
    from urh_amd.pipeline import DemodParams
    import os
    rng = np.random.default_rng(int(os.environ.get("URH_FUZZ_SEED", "2026")) + (7 if big else 0))     # URH_FUZZ_SEED: more rounds by hand
    for it in range(int(os.environ.get("URH_FUZZ_CASES", "160")) if cases is None else cases):
        n = int(rng.choice([rng.integers(3, 300), rng.integers(300, 9000), rng.integers(9000, 70_000)]))
        if big:
            n = int(rng.choice([n, rng.integers(70_000, 600_000), 8192 * int(rng.integers(1, 40)) + int(rng.choice([0, 1, 127, 128, 2047, 2048, 4097]))]))
        sps = int(rng.choice([1, 2, 5, 17, 100, 333]))
        dtype = [np.float32, np.float32, np.int8, np.uint8, np.int16, np.uint16][it % 6]
        mod = "FSK" if it % 2 == 0 else "ASK"
        pe = int(rng.choice([0, max(n // 3, 1), 2500]))
        dev_hz = (20e3, 20e3, 60e3, 90e3, 140e3, 260e3)[(it // 6) % 6]     # phase steps 0.13 ... 1.6 rad per sample
        iq = synth_fsk(n, sps=sps, seed=it, noise=float(rng.choice([0.0, 0.02, 0.3])), pause_every=pe,
                       pause_len=int(rng.choice([1, 7, 130, 2100])) if pe else 0, dtype=dtype, deviation_hz=dev_hz)
        scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64)).max())
        if mod == "ASK":
            env = np.repeat(rng.integers(0, 2, n // sps + 1), sps)[:n]
            iq = (iq.astype(np.float32) * (0.05 + 0.95 * env)[:, None]).astype(dtype)
        tol = int(rng.choice([0, 1, 2, 5, 9, 64, 200]))
        noise = float(rng.choice([0.0, 0.2, 0.6])) * scale
        center = float(rng.choice([0.0, 0.1, -0.2])) if mod == "FSK" else float(rng.choice([0.05, 0.35, 0.8]))
        pt = int(rng.choice([0, 1, 8]))
        bps, spacing = (2, float(rng.choice([0.05, 0.3]))) if it % 5 == 3 else (1, 1.0)     # order 4: three thresholds
        p = DemodParams(mod, bps, noise, center, spacing, tol, sps, 0.1, pt, True)
        qad = oracle.afp_demod(iq, noise, mod, 2)
        pp = oracle.grab_pulse_lens(qad, center, tol, mod, sps, bps, spacing)
        fb = oracle.ppseq_to_bits_flat(pp, sps, bps, True, pt)
        res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // (tol + 1) + 2)
        ctxt = (it, n, sps, np.dtype(dtype).name, mod, tol, noise, center, pt, bps, spacing)
        assert bits_equal(res.qad.cpu().numpy(), qad), ctxt
        assert np.array_equal(res.ppseq(), pp), ctxt
        assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat())), ctxt


@pytest.mark.parametrize("tiles", [2, 4])
def test_forced_chunk_plan_vs_oracle(pipe, oracle, tiles):
    """The 1 GiB benchmark runs chunks of 4 tiles = 64 rows (every lane of the run phase populated, four wavefronts per chunk
    exchanging bit planes through LDS); captures the oracle finishes in seconds would get 1-tile chunks.  Forcing the plan
    (urhgpu_test_force_tiles_per_chunk) puts the benchmark's kernel configuration under the full differential suite: the
    randomised fuzz (all dtypes, orders 2 and 4, tolerances, noise gates), the medium captures, the order-4 comparison and the
    bit-plane / state-byte comparison -- all against the oracle, bit-exact."""
    from urh_amd import _lib
    lib = _lib.load()
    assert lib.urhgpu_test_force_tiles_per_chunk(tiles) == 0
    try:
        test_fused_randomised_vs_oracle(pipe, oracle, cases=150, big=True)
        for mod in ("FSK", "ASK"):
            test_fused_equals_oracle_medium(pipe, oracle, mod)
            test_bit_plane_kernel_order4_equals_state_byte_kernel(pipe, oracle, mod)
            test_bit_plane_kernel_equals_state_byte_kernel(pipe, oracle, mod)
        test_integer_capture_exact_zero_cross_products(pipe, oracle, np.int8)
        test_wide_deviation_reduced_argument_path(pipe, oracle, np.float32)
        test_wide_loop_every_angle_and_mode_switches(pipe, oracle, np.float32)
    finally:
        lib.urhgpu_test_force_tiles_per_chunk(0)
    assert lib.urhgpu_test_force_tiles_per_chunk(5) == _lib.ERR_ARG


def _four_level(n, sps, seed, mod, noise=0.03):
    """4-FSK (four frequency steps) or 4-ASK (four amplitudes) capture with silent gaps, float32 [n, 2]."""
    rng = np.random.default_rng(seed)
    sym = np.repeat(rng.integers(0, 4, n // sps + 1), sps)[:n]
    if mod == "FSK":
        ph = np.cumsum(np.array([-0.6, -0.2, 0.2, 0.6])[sym])
        amp = np.ones(n)
    else:
        ph = 0.01 * np.arange(n)
        amp = np.array([0.2, 0.45, 0.7, 0.95])[sym]
    iq = np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1) + noise * rng.standard_normal((n, 2))
    for a in range(n // 5, n, n // 4):
        iq[a:a + n // 21] *= 0.01
    return iq.astype(np.float32)


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
def test_bit_plane_kernel_order4_equals_state_byte_kernel(pipe, oracle, mod):
    """Modulation order 4 (three thresholds, states 1..4 as two bit planes + PAUSE) on the bit-plane kernel vs the state-byte
    kernel vs the oracle: four-level captures, tolerances either side of the plane kernel's limit, noise gate on and off."""
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams
    lib = _lib.load()
    center, spacing = (0.0, 0.4) if mod == "FSK" else (0.4, 0.17)          # ASK magnitudes are scaled by 1 / sqrt(2)
    try:
        for n, sps in ((1 << 18, 40), (262_144 + 8192 + 2048 + 130, 25), (70_001, 100), (777, 5)):
            iq = _four_level(n, sps, n % 83, mod)
            iq[np.random.default_rng(n).integers(0, n, 50)] = 0.0
            dev = torch.from_numpy(iq).cuda()
            for tol in (0, 1, 5, 6, 33, 64, 65):
                for noise in (0.0, 0.1):
                    p = DemodParams(mod, 2, noise, center, spacing, tol, sps, 0.1, 8, True)
                    got = []
                    for force in (0, 1):
                        lib.urhgpu_test_force_state_bytes(force)
                        res = pipe.iq_to_bits(dev, p, want_qad=True, cap_rows=n // (tol + 1) + 2)
                        got.append((res.qad.cpu().numpy().copy(), res.ppseq().copy()) + tuple(x.copy() for x in res.flat()))
                    assert bits_equal(got[0][0], got[1][0]), (n, tol, noise)
                    for k in range(1, len(got[0])):
                        assert np.array_equal(got[0][k], got[1][k]), (n, tol, noise, k)
                    if tol in (0, 5, 64):
                        qad = oracle.afp_demod(iq, noise, mod, 4)
                        pp = oracle.grab_pulse_lens(qad, center, tol, mod, sps, 2, spacing)
                        fb = oracle.ppseq_to_bits_flat(pp, sps, 2, True, 8)
                        assert bits_equal(got[0][0], qad) and np.array_equal(got[0][1], pp), (n, tol, noise)
                        assert n < 1000 or len(set(pp[:, 0].tolist())) >= 4, "all four levels should occur"
                        assert all(np.array_equal(a, b) for a, b in zip(fb, got[0][2:])), (n, tol, noise)
    finally:
        lib.urhgpu_test_force_state_bytes(0)


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
def test_bit_plane_kernel_equals_state_byte_kernel(pipe, oracle, mod):
    """Modulation orders 2 and 4 run k_demod_runs_bp (bit planes, wavefronts sharing a chunk), every other order the
    state-byte kernel: both on the same captures -- tolerances 0..64 and beyond (65: the state-byte kernel is the
    only one), noise gating on and off, exact-zero samples, sizes that end in whole rows, partial rows, partial tiles --
    must agree on everything, and with the oracle."""
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams
    lib = _lib.load()
    rng = np.random.default_rng(77 + len(mod))
    try:
        for n, dtype in ((1 << 18, np.float32), (262_144 + 8192 + 2048 + 130, np.float32), (70_001, np.int16), (8192 * 3, np.uint8)):
            iq = synth_fsk(n, sps=50, seed=n % 89, noise=0.08, pause_every=n // 4, pause_len=n // 19, dtype=dtype)
            scale = {np.float32: 1.0, np.int16: 32767.0, np.uint8: 127.0}[dtype]
            if mod == "ASK":
                env = np.repeat(rng.integers(0, 2, n // 50 + 1), 50)[:n]
                iq = (iq.astype(np.float32) * (0.05 + 0.95 * env)[:, None]).astype(dtype)
            if dtype == np.float32:
                iq[rng.integers(0, n, 300)] = 0.0                       # exact zeros: NOISE even at threshold 0
                iq[5000:5700] = 0.0
            dev = torch.from_numpy(iq).cuda()
            for tol in (0, 1, 2, 5, 6, 17, 63, 64, 65):
                for noise in (0.0, 0.25 * scale):
                    center = 0.0 if mod == "FSK" else 0.35
                    p = DemodParams(mod, 1, noise, center, 1.0, tol, 50, 0.1, 8, True)
                    got = []
                    for force in (0, 1):
                        lib.urhgpu_test_force_state_bytes(force)
                        res = pipe.iq_to_bits(dev, p, want_qad=True, cap_rows=n // (tol + 1) + 2)
                        got.append((res.qad.cpu().numpy().copy(), res.ppseq().copy()) + tuple(x.copy() for x in res.flat()))
                    assert bits_equal(got[0][0], got[1][0]), (n, dtype, tol, noise)
                    for k in range(1, len(got[0])):
                        assert np.array_equal(got[0][k], got[1][k]), (n, dtype, tol, noise, k)
                    if tol in (0, 5, 64) and n <= (1 << 18):
                        qad = oracle.afp_demod(iq, noise, mod, 2)
                        pp = oracle.grab_pulse_lens(qad, center, tol, mod, 50, 1, 1.0)
                        assert bits_equal(got[0][0], qad) and np.array_equal(got[0][1], pp), (n, dtype, tol, noise)
    finally:
        lib.urhgpu_test_force_state_bytes(0)


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_single_gpu(pipe, oracle, mod, world):
    """W simulated ranks (threads, one context each) on this GPU: the stitched sharded result equals the
    single-GPU result and the oracle, bit for bit (qad included), for boundaries that cut runs, pauses and
    messages at arbitrary places."""
    import torch
    from test_sharding import run_threads
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import shard_bounds, stitch
    rng = np.random.default_rng(world * 7 + len(mod))
    for n, sps, dtype in ((300_007, 100, np.float32), (1 << 20, 100, np.float32), (99_999, 20, np.int8)):
        iq = synth_fsk(n, sps=sps, seed=n % 97, noise=0.05, pause_every=n // 5, pause_len=n // 23, dtype=dtype)
        scale = 1.0 if dtype == np.float32 else 127.0
        if mod == "ASK":
            env = np.repeat(rng.integers(0, 2, n // sps + 1), sps)[:n]
            iq = (iq.astype(np.float32) * (0.05 + 0.95 * env)[:, None]).astype(dtype)
        noise, tol = 0.2 * scale, 3
        center = 0.0 if mod == "FSK" else 0.35
        p = DemodParams(mod, 1, noise, center, 1.0, tol, sps, 0.1, 8, True)
        dev_iq = torch.from_numpy(iq).cuda()
        single = pipe.iq_to_bits(dev_iq, p, want_qad=True)
        want = (single.ppseq(),) + tuple(single.flat())
        want_qad = single.qad.cpu().numpy().copy()
        if n <= 300_007:
            qad = oracle.afp_demod(iq, noise, mod, 2)
            pp = oracle.grab_pulse_lens(qad, center, tol, mod, sps, 1, 1.0)
            assert np.array_equal(want[0], pp) and bits_equal(want_qad, qad)
        cuts = [0] + sorted(int(c) * 8 for c in rng.choice(np.arange(1, n // 8), size=world - 1, replace=False)) + [n]
        for bounds in (shard_bounds(n, world), [(cuts[r], cuts[r + 1]) for r in range(world)]):
            shards = [dev_iq[a:b] for a, b in bounds]
            # the halo through the first exchange, and handed over with the shard (urhgpu_shard_launch_dev: no halo exchange)
            for halos in (None, [None] + [dev_iq[a - 2:a].clone() for a, _ in bounds[1:]]):
                res = run_threads(world, lambda r: GpuShardEngine(0), shards, bounds, n, p, halos)
                got = stitch(res)
                for k, (a, b) in enumerate(zip(got, want)):
                    assert np.array_equal(a, b), (mod, world, n, bounds, halos is not None, k, len(a), len(b))
                got_qad = np.concatenate([r.qad.cpu().numpy() for r in res])
                assert bits_equal(got_qad, want_qad)


def test_shard_summary_one_launch_equals_generic(pipe):
    """the local pass of a sharded capture as ONE launch (k_shard_summary: the shard's chunks composed as ResElems) delivers the same 72
    summary bytes as the three generic resolve launches (tuning key shard_summary_generic), for shards cut anywhere: inside runs, pauses
    and messages, shards of a few samples, shards without a stable run, tolerances that leave trailing runs open at the shard end
    (signal_functions.pyx:421-495 across a boundary)"""
    import torch
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    one, gen = GpuShardEngine(0), GpuShardEngine(0, tuning={"shard_summary_generic": 1})
    rng = np.random.default_rng(20260930)
    seen_open = seen_empty = 0
    for it in range(60):
        n = int(rng.choice([int(rng.integers(64, 5000)), int(rng.integers(5000, 400_000)), int(rng.integers(400_000, 3 << 20))]))
        n -= n % 8
        sps = int(rng.choice([5, 20, 100, 333]))
        dtype = [np.float32, np.int16, np.int8][int(rng.integers(0, 3))]
        mod = "FSK" if rng.random() < 0.6 else "ASK"
        iq = synth_fsk(n, sps=sps, seed=it, noise=0.05, pause_every=max(n // 5, 8), pause_len=max(n // 23, 1), dtype=dtype)
        scale = 1.0 if dtype == np.float32 else (127.0 if dtype == np.int8 else 8192.0)
        if mod == "ASK":
            env = np.repeat(rng.integers(0, 2, n // sps + 1), sps)[:n]
            iq = (iq.astype(np.float32) * (0.05 + 0.95 * env)[:, None]).astype(dtype)
        tol = int(rng.choice([1, 3, 5, 40, 64, 200]))
        p = DemodParams(mod, 1, 0.2 * scale, 0.0 if mod == "FSK" else 0.35, 1.0, tol, sps, 0.1, 8, False)
        dev_iq = torch.from_numpy(iq).cuda()
        world = int(rng.integers(2, 9))
        cuts = [0] + sorted(int(c) * 8 for c in rng.choice(np.arange(1, n // 8), size=world - 1, replace=False)) + [n]
        for r in range(world):
            a, b = cuts[r], cuts[r + 1]
            left = dev_iq[a - 2:a].clone() if r else None
            got = []
            for e in (one, gen):
                got.append(e.runs(dev_iq[a:b], left, a, n, r, world, p, False).cpu().numpy().copy())
                _lib_abort(e)
            # (68 of the 72 bytes are fields; the last four are the struct's padding)
            assert np.array_equal(got[0].view(np.uint8)[:68], got[1].view(np.uint8)[:68]), (it, mod, n, sps, tol, dtype, r, world, got[0], got[1])
            seen_open += int(got[0][0] >= 0)
            seen_empty += int((got[0][5] & 0xFFFFFFFF) == 0)
    assert seen_open > 0 and seen_empty > 0              # summaries with an open trailing run / without an accepted run were among them


def _lib_abort(engine):
    """drop a sharded pass behind its local pass (the next urhgpu_shard_runs_dev starts a new one)"""
    import torch
    torch.cuda.synchronize()
    engine._res = None
    engine._keep = ()


def test_sharded_fuzz_extended(pipe, oracle):
    """long form of the sharded check (URH_FUZZ_ROUNDS rounds, default 6; URH_FUZZ_SEED): random world (2 .. 12), shard boundaries
    anywhere (multiples of 8 samples; shards of a few samples, shards inside a pause, shards without a single run), FSK orders 2 / 4 and
    ASK, float32 / int16 / int8 captures, halos exchanged or handed over: the stitched pieces equal ONE single-GPU pass (itself compared
    with the oracle on the shorter captures), bit for bit, qad included (signal_functions.pyx:333-495 across shard boundaries)."""
    import os
    import torch
    from test_sharding import run_threads
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import stitch
    rounds = int(os.environ.get("URH_FUZZ_ROUNDS", "6"))
    seed0 = int(os.environ.get("URH_FUZZ_SEED", "0"))
    for it in range(rounds):
        rng = np.random.default_rng([991, seed0, it])
        world = int(rng.integers(2, 13))
        n = int(rng.choice([int(rng.integers(8 * world + 8, 5000)), int(rng.integers(5000, 400_000)), int(rng.integers(400_000, 1 << 21))]))
        n -= n % 8
        sps = int(rng.choice([5, 20, 100, 333]))
        dtype = [np.float32, np.int16, np.int8][int(rng.integers(0, 3))]
        mod = "FSK" if rng.random() < 0.6 else "ASK"
        bps, spacing = ((2, float(rng.choice([0.05, 0.3]))) if (mod == "FSK" and rng.random() < 0.3) else (1, 1.0))
        tol = int(rng.choice([0, 1, 3, 9, 40]))
        pe = int(rng.choice([0, max(n // 5, 1), 2500]))
        iq = synth_fsk(n, sps=sps, seed=int(rng.integers(0, 1 << 30)), noise=float(rng.choice([0.0, 0.05, 0.3])), pause_every=pe,
                       pause_len=int(rng.choice([7, 130, max(n // 23, 1)])), dtype=dtype)
        scale = 1.0 if dtype == np.float32 else float(np.iinfo(dtype).max) * 0.7
        if mod == "ASK":
            env = np.repeat(rng.integers(0, 2, n // sps + 1), sps)[:n]
            iq = (iq.astype(np.float32) * (0.05 + 0.95 * env)[:, None]).astype(dtype)
            noise, center = float(rng.choice([0.0, 0.2])) * scale, float(rng.choice([0.35, 0.5])) * scale
        else:
            noise, center = float(rng.choice([0.0, 0.2])) * scale, float(rng.choice([0.0, 0.1, -0.2]))
        p = DemodParams(mod, bps, noise, center, spacing, tol, sps, 0.1, int(rng.choice([0, 1, 8])), True)
        dev_iq = torch.from_numpy(iq).cuda()
        single = pipe.iq_to_bits_checked(dev_iq, p, want_qad=True)
        want = (single.ppseq(),) + tuple(single.flat())
        want_qad = single.qad.cpu().numpy().copy()
        if n <= 400_000:
            qad = oracle.afp_demod(iq, noise, mod, 2 ** bps)
            pp = oracle.grab_pulse_lens(qad, center, tol, mod, sps, bps, spacing)
            assert np.array_equal(want[0], pp) and bits_equal(want_qad, qad), (it, seed0)
        cuts = [0] + sorted(int(c) * 8 for c in rng.choice(np.arange(1, n // 8), size=world - 1, replace=False)) + [n]
        bounds = [(cuts[r], cuts[r + 1]) for r in range(world)]
        shards = [dev_iq[a:b] for a, b in bounds]
        halos = None if rng.random() < 0.5 else [None] + [dev_iq[a - 2:a].clone() for a, _ in bounds[1:]]
        tag = (it, seed0, mod, 2 ** bps, np.dtype(dtype).name, world, n, sps, tol, bounds, halos is not None)
        res = run_threads(world, lambda r: GpuShardEngine(0, worst_case_rows=True), shards, bounds, n, p, halos)
        got = stitch(res)
        for k, (a, b) in enumerate(zip(got, want)):
            assert np.array_equal(a, b), (tag, k, len(a), len(b))
        got_qad = np.concatenate([r.qad.cpu().numpy() for r in res])
        assert bits_equal(got_qad, want_qad), tag


# ---- filters, magnitudes, noise estimator ------------------------------------------------------------------
def cbits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.complex64).view(np.uint32), np.ascontiguousarray(b, np.complex64).view(np.uint32)
    fa, fb = a.view(np.float32), b.view(np.float32)
    return bool(((a == b) | (np.isnan(fa) & np.isnan(fb))).all())


def test_fir_kat(sf):
    """/root/reference/tests/test_filter.py:20-31"""
    x = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 42], dtype=np.complex64)
    out = sf.fir_filter(x, np.array([0.25] * 4, dtype=np.complex64))
    assert np.allclose(out, [0.25, 0.75, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 16.5])


def test_fir_equals_oracle(sf, oracle):
    rng = np.random.default_rng(12)
    for n in (1, 2, 3, 5, 63, 64, 1023, 1024, 1025, 4097, 50_001):
        for m in (1, 2, 3, 4, 5, 10, 63, 64, 65, 257):
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            h = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
            assert cbits_equal(sf.fir_filter(x, h), oracle.fir_filter(x, h)), (n, m)
    # signed zeros, infinities, NaNs, denormals: the Annex G recovery path of the complex product
    vals = np.array([0.0, -0.0, 1.0, -1.0, 1e-40, 3e38, -3e38, np.inf, -np.inf, np.nan, 0.5], dtype=np.float32)
    x = (vals[rng.integers(0, len(vals), 5000)] + 1j * vals[rng.integers(0, len(vals), 5000)]).astype(np.complex64)
    h = (vals[rng.integers(0, 7, 9)] + 1j * vals[rng.integers(0, 7, 9)]).astype(np.complex64)
    assert cbits_equal(sf.fir_filter(x, h), oracle.fir_filter(x, h))
    assert len(sf.fir_filter(np.zeros(0, np.complex64), h)) == 0
    # a capture whose tiles mostly take the fast kernel while a few are handed to the checked one (non-finite / huge samples), leading
    # negative zeros in front (the zero-history identity), a partial last tile
    for m in (7, 64, 100):
        x = (rng.standard_normal(20_001) + 1j * rng.standard_normal(20_001)).astype(np.complex64)
        x[:5] = np.array([-0.0, 0.0, -0.0 - 0.0j, 1e-30, -0.0], dtype=np.complex64)
        x[7000] = np.inf
        x[15_000] = np.float32(1e25)
        x[19_990] = np.nan
        h = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
        h[m // 2] = np.complex64(-0.0)
        assert cbits_equal(sf.fir_filter(x, h), oracle.fir_filter(x, h)), m


def test_fir_with_left_halo_equals_one_pass(pipe, oracle):
    """sharded FIR: a shard filtered with the m-1 preceding samples as halo equals the same stretch of the one-pass result"""
    import torch
    from urh_amd import _lib
    rng = np.random.default_rng(2)
    n, m, cut = 30_000, 64, 12_345
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    h = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
    want = oracle.fir_filter(x, h)
    dx, dh = torch.from_numpy(x).cuda(), torch.from_numpy(h).cuda()
    out = torch.empty(n - cut, dtype=torch.complex64, device="cuda")
    halo = dx[cut - (m - 1):cut].clone()
    shard = dx[cut:].clone()
    pipe.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().urhgpu_fir_filter_dev(pipe.ctx.handle, C.c_void_p(shard.data_ptr()), n - cut, C.c_void_p(dh.data_ptr()), m,
                                                 C.c_void_p(halo.data_ptr()), C.c_void_p(out.data_ptr())))
    assert cbits_equal(out.cpu().numpy(), want[cut:])


def test_iir_equals_oracle(sf, oracle):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    for na, nb in ((1, 0), (3, 2), (2, 4), (5, 5)):
        a, b = rng.standard_normal(na) * 0.3, rng.standard_normal(nb) * 0.2
        assert cbits_equal(sf.iir_filter(a, b, x), oracle.iir_filter(a, b, x)), (na, nb)


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.float32])
def test_integer_capture_exact_zero_cross_products(pipe, oracle, dtype):
    """Integer samples make the FSK cross product I0*Q1 - Q0*I1 an exact integer that is often exactly zero (and, with zero
    components, a zero whose sign depends on how the reference multiplies): small amplitudes, so that zeros of every kind
    are frequent; qad bit for bit (signed zeros included), pulse table, bits.  float32: the same integers divided by 128, a
    complex64 recording of an 8-bit receiver."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(7)
    zero_frac = 0.0
    for amp, n in ((3, 200_003), (9, 262_144), (40, 300_000)):
        ph = np.cumsum(rng.choice([-0.13, 0.13], n // 50 + 1).repeat(50)[:n])
        x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1) + 0.3 * rng.standard_normal((n, 2))
        off = 128 if dtype == np.uint8 else 0
        if dtype == np.float32:
            iq = (np.clip(np.round(x), -128, 127) / 128.0).astype(np.float32)
        else:
            info = np.iinfo(dtype)
            iq = np.clip(np.round(x) + off, info.min, info.max).astype(dtype)
        iq[1000:1100] = off                                              # a stretch of exact zeros (uint8: of the offset)
        for noise in (0.0, 1.5 / 128 if dtype == np.float32 else 1.5):
            p = DemodParams("FSK", 1, noise, 0.0, 1.0, 2, 50, 0.1, 8, True)
            qad = oracle.afp_demod(iq, noise, "FSK", 2)
            pp = oracle.grab_pulse_lens(qad, 0.0, 2, "FSK", 50, 1, 1.0)
            fb = oracle.ppseq_to_bits_flat(pp, 50, 1, True, 8)
            res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // 3 + 2)
            got = res.qad.cpu().numpy()
            assert bits_equal(got, qad), (amp, noise, int((got.view(np.uint32) != qad.view(np.uint32)).sum()))
            assert np.array_equal(res.ppseq(), pp)
            assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))
            zero_frac = max(zero_frac, float((qad[2:] == 0).mean()))
    assert dtype == np.uint8 or zero_frac > 1e-3                         # the case is actually exercised


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.int8])
def test_wide_deviation_reduced_argument_path(pipe, oracle, dtype):
    """Phase steps around and beyond atan(7/16) per sample (wide FSK deviations, noise): rows whose lanes need fdlibm's first two
    argument reductions (7/16 <= |im/re| < 1) are evaluated without the general atan2f; steps up to pi/4 and beyond (general
    path), mixed in one capture, qad bit for bit against the oracle."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(17)
    n = 400_000
    for step, sigma in ((0.30, 0.05), (0.42, 0.02), (0.55, 0.05), (0.70, 0.03), (0.80, 0.08), (1.30, 0.05)):
        ph = np.cumsum(rng.choice([-step, step], n // 40 + 1).repeat(40)[:n])
        x = np.stack([np.cos(ph), np.sin(ph)], 1) + sigma * rng.standard_normal((n, 2))
        x[100_000:100_500] *= 0.0                                        # exact zeros
        if dtype == np.float32:
            iq = x.astype(np.float32)
            iq[200_000:200_064, 0] = 1.0; iq[200_000:200_064, 1] = np.linspace(-1.2, 1.2, 64, dtype=np.float32)   # |im/re| sweeps
        else:
            amp = 0.6 * np.iinfo(dtype).max
            iq = np.clip(np.round(x * amp), np.iinfo(dtype).min, np.iinfo(dtype).max).astype(dtype)
        for noise in (0.0, 0.3 if dtype == np.float32 else 0.3 * 0.6 * np.iinfo(dtype).max):
            p = DemodParams("FSK", 1, noise, 0.0, 1.0, 3, 40, 0.1, 8, True)
            qad = oracle.afp_demod(iq, noise, "FSK", 2)
            pp = oracle.grab_pulse_lens(qad, 0.0, 3, "FSK", 40, 1, 1.0)
            res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // 4 + 2)
            got = res.qad.cpu().numpy()
            assert bits_equal(got, qad), (step, sigma, noise, int((got.view(np.uint32) != qad.view(np.uint32)).sum()))
            assert np.array_equal(res.ppseq(), pp), (step, sigma, noise)


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.int8, np.uint8])
def test_wide_loop_every_angle_and_mode_switches(pipe, oracle, dtype):
    """The batch-level wide loop of the hot kernel (fsk_wide: fdlibm's argument reduction for a whole batch, entered by a batch the fast
    loop flags, left after four batches that one would have taken or by a batch outside its window): phase steps in every range of
    atanf's reduction and beyond pi/2 (re < 0: pi - (z - pi_lo)), stretches of narrow and wide steps in turn (the loops hand over in
    both directions), pauses and exact zeros inside wide stretches (the wide window fails: generic step, then back), and -- integer
    samples -- exactly zero cross products with re < 0 (atan2f(+-0, re < 0) = +-pi with the sign of the reference's product)."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(29)
    n = 600_000
    steps = np.concatenate([rng.choice([-s, s], 1500) for s in (0.1, 0.45, 0.1, 0.6, 0.9, 0.2, 1.4, 1.7, 0.05, 2.2, 2.9, 3.1, 0.3, 1.0)])
    ph = np.cumsum(np.resize(steps.repeat(40), n))
    amp = np.ones(n)
    amp[150_000:150_700] = 0.0                                           # exact zeros inside a wide stretch
    amp[330_000:333_000] = 0.02                                          # a pause below the noise threshold of the second pass
    x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1) + 0.02 * rng.standard_normal((n, 2)) * (amp[:, None] > 0)
    if dtype == np.float32:
        iq = x.astype(np.float32)
        iq[400_000:400_256] = np.tile(np.array([[0.5, 0.25], [-0.5, -0.25]], np.float32), (128, 1))     # im == 0 exactly, re < 0: +-pi
        scale = 1.0
    else:
        info = np.iinfo(dtype)
        scale = 0.3 * (info.max - info.min) / 2 if dtype != np.int8 else 9.0     # int8 at a small amplitude: many exactly zero cross products
        mid = (info.max + info.min + 1) // 2 if dtype == np.uint8 else 0
        iq = np.clip(np.round(x * scale) + mid, info.min, info.max).astype(dtype)
        k = np.array([[5, 3], [-5, -3], [-10, -6], [10, 6], [0, 7], [0, -7]])
        iq[400_000:400_000 + 6 * 40] = (np.tile(k, (40, 1)) + mid).astype(dtype)
    zero_back = 0
    for noise in (0.0, 0.1 * scale):
        p = DemodParams("FSK", 1, noise, 0.0, 1.0, 3, 40, 0.1, 8, True)
        qad = oracle.afp_demod(iq, noise, "FSK", 2)
        pp = oracle.grab_pulse_lens(qad, 0.0, 3, "FSK", 40, 1, 1.0)
        res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // 4 + 2)
        got = res.qad.cpu().numpy()
        assert bits_equal(got, qad), (noise, int((got.view(np.uint32) != qad.view(np.uint32)).sum()), np.nonzero(got.view(np.uint32) != qad.view(np.uint32))[0][:8])
        assert np.array_equal(res.ppseq(), pp), noise
        zero_back = max(zero_back, int((np.abs(qad) == np.float32(np.pi)).sum()))
    assert dtype == np.uint8 or zero_back > 50                           # +-pi from exactly zero cross products behind: exercised (unsigned samples are not centred: re > 0)


@pytest.mark.parametrize("dtype", [np.int8, np.int16])
@pytest.mark.parametrize("bps", [1, 2])
def test_wide_int_key_one_shot_and_orders(oracle, dtype, bps):
    """tuning key wide_int: a ONE-SHOT pass over a signed integer FSK capture takes the hot kernel's instantiation with the wide loop
    (capture streams pick it by their probe; a one-shot pass has the caller's word only) -- 2-FSK and 4-FSK (the two-plane instantiation),
    narrow and wide deviations, a gated pause and exactly zero samples: demodulated signal bit for bit, pulse table and bits equal the
    reference's whichever instantiation ran (signal_functions.pyx:363-376, 421-495)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = (1 << 19) + 8192 * 3 + 1032
    amp = 0.6 * np.iinfo(dtype).max
    rng = np.random.default_rng(77 + bps)
    plain, wide = DevicePipeline(0), DevicePipeline(0, tuning={"wide_int": 1})
    from urh_amd import _lib
    launches0 = _lib.load().urhgpu_test_wide_int_launches()
    for dev_hz in (15e3, 120e3, 260e3):
        x = synth_fsk(n, sps=40, seed=int(dev_hz) % 89 + bps, noise=0.03, pause_every=n // 3, pause_len=3000, deviation_hz=dev_hz)
        if bps == 2:                                          # four tones: every second symbol at a third of the deviation
            ph = np.unwrap(np.angle(x[:, 0] + 1j * x[:, 1]))
            d = np.diff(ph, prepend=ph[0]) * np.repeat(rng.choice([1.0, 1.0 / 3.0], n // 40 + 1), 40)[:n]
            mag = np.hypot(x[:, 0], x[:, 1])
            x = np.stack([mag * np.cos(np.cumsum(d)), mag * np.sin(np.cumsum(d))], 1)
        iq = np.clip(np.round(x * amp), np.iinfo(dtype).min, np.iinfo(dtype).max).astype(dtype)
        iq[200_000:200_064] = 0
        step = 2 * np.pi * dev_hz / 1e6
        center, spacing = (0.0, 1.0) if bps == 1 else (0.0, 2.0 * step / 3.0)
        p = DemodParams("FSK", bps, 0.1 * amp, center, spacing, 3, 40, 0.1, 8, True)
        qad = oracle.afp_demod(iq, p.noise_threshold, "FSK", 1 << bps)
        pp = oracle.grab_pulse_lens(qad, center, 3, "FSK", 40, bps, spacing)
        fb = oracle.ppseq_to_bits_flat(pp, 40, bps, True, 8)
        for pl in (plain, wide):
            res = pl.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n // 4 + 2)
            got = res.qad.cpu().numpy()
            assert bits_equal(got, qad), (dev_hz, pl is wide, int((got.view(np.uint32) != qad.view(np.uint32)).sum()))
            assert np.array_equal(res.ppseq(), pp), (dev_hz, pl is wide)
            assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat())), (dev_hz, pl is wide)
    assert _lib.load().urhgpu_test_wide_int_launches() - launches0 == 3       # the keyed pipeline's passes took that instantiation, the other's did not


@pytest.mark.parametrize("bps", [1, 2])
def test_many_huge_rows_expand(pipe, oracle, bps):
    """300 constant stretches of 4 500-9 000 symbols each (rows of more than 4096 bits go to k_expand_huge's work list, more
    rows than its grid has row slots), one sample per symbol, short breaks in between: bits and bit_sample_pos vs the oracle."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(90 + bps)
    amp = np.concatenate([np.concatenate([np.full(int(rng.integers(4500, 9000)), 0.8 if bps == 1 else float(rng.choice([0.55, 0.8]))),
                                          np.full(int(rng.integers(1, 4)), 0.05)]) for _ in range(300)])
    n = len(amp)
    ph = 0.01 * np.arange(n)
    iq = (np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1) + 0.002 * rng.standard_normal((n, 2))).astype(np.float32)
    center, spacing = (0.25, 1.0) if bps == 1 else (0.3, 0.17)          # ASK magnitudes are scaled by 1 / sqrt(2)
    p = DemodParams("ASK", bps, 0.0, center, spacing, 0, 1, 0.1, 8, True)
    qad = oracle.afp_demod(iq, 0.0, "ASK", 1 << bps)
    pp = oracle.grab_pulse_lens(qad, center, 0, "ASK", 1, bps, spacing)
    assert (pp[:, 1] > 4096).sum() >= 290
    fb = oracle.ppseq_to_bits_flat(pp, 1, bps, True, 8)
    res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True, cap_rows=n + 2)
    assert np.array_equal(res.ppseq(), pp)
    assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))


@pytest.mark.parametrize("bps,stretches", [(1, 300), (2, 300), (1, 9000)])
def test_tile_tail_huge_rows_fsk(pipe, oracle, bps, stretches):
    """The tile tail's own huge-row path (rows of more than 4096 bits are listed while the rows are written and expanded by extra
    workgroups of the expansion launch): constant-frequency stretches of 4 500-9 000 symbols at one sample per symbol, 2-FSK and
    4-FSK, against the oracle; 9 000 stretches overflow the 8 192-entry list (every wavefront then expands its own rows)."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(190 + bps + stretches)
    lo, hi = (4500, 9000) if stretches < 1000 else (4200, 4700)
    steps = [-0.5, 0.5] if bps == 1 else [-0.6, -0.2, 0.2, 0.6]
    L = len(steps)
    f = np.concatenate([np.concatenate([np.full(int(rng.integers(lo, hi)), steps[k % L]), np.full(int(rng.integers(1, 4)), steps[(k + L // 2 + (L == 2)) % L])])
                        for k in range(stretches)])
    n = len(f)
    ph = np.cumsum(f)
    iq = (np.stack([np.cos(ph), np.sin(ph)], 1) + 0.002 * rng.standard_normal((n, 2))).astype(np.float32)
    center, spacing = (0.0, 1.0) if bps == 1 else (0.0, 0.4)
    p = DemodParams("FSK", bps, 0.0, center, spacing, 0, 1, 0.1, 8, True)
    qad = oracle.afp_demod(iq, 0.0, "FSK", 1 << bps)
    pp = oracle.grab_pulse_lens(qad, center, 0, "FSK", 1, bps, spacing)
    assert (pp[:, 1] * bps > 4096).sum() >= 0.95 * stretches
    fb = oracle.ppseq_to_bits_flat(pp, 1, bps, True, 8)
    dev = torch.from_numpy(iq).cuda()
    for _ in range(3):                              # consecutive passes alternate the two huge-row counters
        res = pipe.iq_to_bits(dev, p, want_qad=True, cap_rows=n + 2)
        assert np.array_equal(res.ppseq(), pp)
        assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))
    # a rows-only pass in between must not disturb the counters
    from urh_amd import signal_functions as sf
    assert np.array_equal(sf.grab_pulse_lens(qad, center, 0, "FSK", 1, bps, spacing), pp)
    res = pipe.iq_to_bits(dev, p, want_qad=False, cap_rows=n + 2)
    assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))


def test_generic_tail_still_equals_oracle(pipe, oracle):
    """Single-GPU FSK captures normally take the tile tail; the generic tail (ASK, sharded captures) stays covered for them too:
    the same fuzz and medium captures with the tile tail switched off."""
    from urh_amd import _lib
    lib = _lib.load()
    lib.urhgpu_test_force_generic_tail(1)
    try:
        test_fused_randomised_vs_oracle(pipe, oracle, cases=120)
        test_fused_equals_oracle_medium(pipe, oracle, "FSK")
    finally:
        lib.urhgpu_test_force_generic_tail(0)


def test_filters_equal_reference_goldens(sf):
    """The real reference's fir_filter / iir_filter outputs (tests/golden/filter/fir_iir.npz) through the C ABI."""
    import os
    g = np.load(os.path.join(ROOT, "tests", "golden", "filter", "fir_iir.npz"))
    for name in (str(n) for n in g["names"]):
        if name.startswith("iir"):
            got = sf.iir_filter(g[name + "_a"], g[name + "_b"], g[name + "_x"])
        else:
            got = sf.fir_filter(g[name + "_x"], g[name + "_h"])
        assert cbits_equal(got, g[name + "_y"]), name


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.uint8, np.int16, np.uint16])
def test_get_magnitudes_equals_oracle(oracle, dtype):
    from urh_amd import util
    for n in (1, 7, 1000, 100_003):
        iq = synth_fsk(n, sps=20, seed=n, noise=0.2, pause_every=300, pause_len=50, dtype=dtype)
        if dtype == np.uint16:
            iq[::7] = 65535                      # C-int overflow -> NaN, as in the reference
        assert np.array_equal(util.get_magnitudes(iq), oracle.get_magnitudes(iq), equal_nan=True), (np.dtype(dtype).name, n)


@pytest.mark.parametrize("name", [c for c in GOLDEN_CASES])
def test_detect_noise_level_golden(pipe, oracle, name):
    import torch
    from urh_amd import estimators
    g = load_golden(name)
    want = oracle.detect_noise_level(oracle.get_magnitudes(g["iq"]))
    got = estimators.detect_noise_level_dev(pipe, torch.from_numpy(g["iq"]).cuda())
    assert got == want, (name, got, want)


def test_detect_noise_level_synthetic(pipe, oracle):
    import torch
    from urh_amd import estimators
    for n, pause_len in ((10, 0), (101, 0), (250_000, 30_000), (1_000_003, 150_000)):
        iq = synth_fsk(n, sps=100, seed=n, noise=0.02, pause_every=max(n // 3, 1), pause_len=pause_len)
        want = oracle.detect_noise_level(oracle.get_magnitudes(iq))
        got = estimators.detect_noise_level_dev(pipe, torch.from_numpy(iq).cuda())
        assert got == want, (n, got, want)


def test_fast_path_division_is_ieee_division():
    """the unscaled Newton/residual division of the FSK fast path returns the bits of the IEEE division"""
    from urh_amd import _lib
    ctx = _lib.Context(0)
    bad = C.c_uint64(123)
    _lib.check(_lib.load().urhgpu_test_fast_division_dev(ctx.handle, 7, 4096, C.byref(bad)))     # 4.3e9 pairs
    assert bad.value == 0


@pytest.mark.parametrize("order", [2, 4])
@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.uint8, np.int16, np.uint16])
def test_costas_equals_oracle(sf, pipe, oracle, order, dtype):
    """PSK: the Costas loop (serial recurrence with glibc sinf / cosf) is bit-exact from sample 1 on; pulse table and
    bits follow (config 5 of BASELINE.json at a size the oracle finishes in seconds)."""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(order * 10 + np.dtype(dtype).itemsize)
    n, sps = 120_000, 100
    sym = rng.integers(0, order, n // sps + 1)
    phases = (np.array([-135, -45, 45, 135]) if order == 4 else np.array([-90, 90]))[sym] * np.pi / 180
    ph = np.repeat(phases, sps)[:n] + 2 * np.pi * 0.04 * np.arange(n)
    iq = np.stack([np.cos(ph), np.sin(ph)], 1) + 0.1 * np.sqrt(0.5) * rng.standard_normal((n, 2))
    iq[40_000:43_000] *= 0.01                                                  # a gap: noise-gated samples freeze the loop
    if dtype == np.float32:
        iq, noise = iq.astype(np.float32), 0.2
    else:
        info = np.iinfo(dtype)
        scale, off = (info.max - info.min) / 2 * 0.7, (info.max + info.min + 1) / 2
        iq = np.clip(np.round(iq * scale + off), info.min, info.max).astype(dtype)
        noise = 0.0 if np.dtype(dtype).kind == "u" else 0.2 * scale               # unsigned: the offset dominates |x|
    want = oracle.afp_demod(iq, noise, "PSK", order, 0.1)
    got = sf.afp_demod(iq, noise, "PSK", order, 0.1)
    assert bits_equal(got[1:], want[1:]), (order, np.dtype(dtype).name, int((got[1:] != want[1:]).sum()))
    assert got[0] == -4.0
    want[0] = -4.0
    bps = 2 if order == 4 else 1
    p = DemodParams("PSK", bps, noise, 0.0, 1.5 if order == 4 else 1.0, 5, sps, 0.1, 8, True)
    pp = oracle.grab_pulse_lens(want, p.center, 5, "PSK", sps, bps, p.center_spacing)
    fb = oracle.ppseq_to_bits_flat(pp, sps, bps, True, 8)
    res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True)
    assert bits_equal(res.qad.cpu().numpy(), want)
    assert np.array_equal(res.ppseq(), pp)
    assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))


# ---- estimators: segmentation, center, plateau lengths (rows 11-13) ----------------------------------------------
def _bursty(n, seed, sps=50):
    rng = np.random.default_rng(seed)
    iq = synth_fsk(n, sps=sps, seed=seed, noise=0.02, pause_every=max(n // 4, 1), pause_len=n // 9)
    # sprinkle short dropouts / spikes so that the 10-sample outlier tolerance matters
    for _ in range(20):
        a = int(rng.integers(0, n))
        ln = int(rng.choice([1, 3, 9, 10, 11, 25]))
        iq[a:a + ln] *= np.float32(rng.choice([0.001, 30.0]))
    return iq


def _check_message_ranges(pipe, oracle, iq, nt, want_segments, caps=(4096, 65536)):
    """the device-resident segmentation + OOK merge against the oracle's segments and the reference's merge of them"""
    import torch
    from urh_amd import estimators
    want = [(int(a), int(b)) for a, b in want_segments]
    seg, n_seg, mrg, n_mrg, amb = estimators.message_ranges_dev(pipe, torch.from_numpy(iq).cuda(), nt, cap_seg=caps[0], cap_merged=caps[1])
    assert n_seg == len(want) and [tuple(r) for r in seg.tolist()] == want[:caps[0]], (len(iq), nt, n_seg, len(want), seg[:3], want[:3])
    want_m = oracle.merge_message_segments_for_ook(list(want))
    if not amb:
        assert n_mrg == len(want_m) and [tuple(r) for r in mrg.tolist()] == [(int(a), int(b)) for a, b in want_m][:caps[1]], \
            (len(iq), nt, n_mrg, len(want_m), mrg[:3], want_m[:3])
    seg2, n2, none, _, _ = estimators.message_ranges_dev(pipe, torch.from_numpy(iq).cuda(), nt, merge=False, cap_seg=3)
    assert none is None and n2 == len(want) and [tuple(r) for r in seg2.tolist()] == want[:3]
    return amb


def test_message_ranges_beyond_the_scan_grid(pipe, oracle):
    """more than 2^20 segments: the segment / cut scans (msg_ranges.hip) loop over their tiles on a bounded grid (scan.hpp scan_grid)"""
    n = 1 << 25
    rng = np.random.default_rng(77)
    env = ((np.arange(n) % 28) < 14).astype(np.float32)
    iq = np.ascontiguousarray((env[:, None] * np.array([0.6, 0.5], np.float32) + 0.01 * rng.standard_normal((n, 2))).astype(np.float32))
    want = oracle.segment_messages_from_magnitudes(oracle.get_magnitudes(iq), 0.2)
    assert len(want) > (1 << 20)
    _check_message_ranges(pipe, oracle, iq, 0.2, want, caps=(1 << 21, 1 << 21))


def test_message_ranges_ook_bursts(pipe, oracle):
    """OOK captures: one segment per pulse, merged into bursts on the device (AutoInterpretation.py:107-148)"""
    def synth_ook_bursts(n, sps, seed):
        rng = np.random.default_rng(seed)
        env = np.zeros(n, np.float32)
        pos = int(rng.integers(0, 5 * sps))
        while pos < n:
            bits = rng.integers(0, 2, int(rng.integers(16, 200)))
            bits[0] = 1
            sym = np.repeat(bits, sps).astype(np.float32)
            # pulse-width jitter: pulses of 1..3 symbols whose lengths spread around the mean
            env[pos:pos + len(sym)] = sym[:max(0, n - pos)]
            pos += len(sym) + int(rng.integers(20, 400)) * sps
        ph = rng.uniform(0, 2 * np.pi)
        t = np.arange(n, dtype=np.float64)
        c = np.exp(1j * (2 * np.pi * 0.01 * t + ph))
        iq = (env * c).astype(np.complex64) + (0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        return np.ascontiguousarray(iq.view(np.float32).reshape(-1, 2))
    n_amb = 0
    for n, seed, sps in ((40_000, 1, 20), (300_000, 2, 50), (2_000_000, 3, 25), (1_500_000, 4, 100)):
        iq = synth_ook_bursts(n, sps=sps, seed=seed)
        for nt in (0.1, 0.3):
            want = oracle.segment_messages_from_magnitudes(oracle.get_magnitudes(iq), nt)
            assert len(want) > 10
            n_amb += _check_message_ranges(pipe, oracle, iq, nt, want, caps=(100, 50_000))
    assert n_amb == 0                                   # the test captures are not borderline: the device decided all of them


def test_segmentation_pass_leaves_the_ask_demodulation(pipe, oracle):
    """urhgpu_message_ranges_demod_dev: the segmentation pass's by-product equals afp_demod(iq, noise, "ASK") bit for bit (oracle and the
    library's own afp_demod), and its ranges equal the plain pass's -- capture lengths around the tile / chunk sizes, a threshold
    that gates many samples and one that gates none"""
    import torch
    from urh_amd import estimators
    from urh_amd.pipeline import DemodParams
    for n, seed in ((1, 1), (2, 2), (2047, 3), (2048, 4), (2049, 5), (8192 * 3, 6), (300_001, 7), (1_234_567, 8)):
        rng = np.random.default_rng(seed)
        env = np.repeat(rng.integers(0, 2, n // 20 + 1), 20)[:n] * (rng.random(n // 2000 + 1) > 0.4).repeat(2000)[:n]
        c = (env * np.exp(2j * np.pi * 0.013 * np.arange(n))) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        iq = np.ascontiguousarray(c.astype(np.complex64).view(np.float32).reshape(-1, 2))
        dev = torch.from_numpy(iq).cuda()
        for nt in (0.0, 0.1, 0.7):
            qad = torch.full((n,), 7.0, dtype=torch.float32, device="cuda")
            got = estimators.message_ranges_dev(pipe, dev, nt, qad_ask=qad)
            plain = estimators.message_ranges_dev(pipe, dev, nt)
            assert np.array_equal(got[0], plain[0]) and got[1] == plain[1] and np.array_equal(got[2], plain[2]) and got[3:] == plain[3:], (n, nt)
            want = oracle.afp_demod(iq, nt, "ASK", 2)
            assert bits_equal(qad.cpu().numpy(), want), (n, nt)
            assert bits_equal(pipe.afp_demod(dev, DemodParams("ASK", 1, nt)).cpu().numpy(), want), (n, nt)
    # integer captures keep the two-pass form
    from urh_amd import _lib
    i8 = torch.zeros((4096, 2), dtype=torch.int8, device="cuda")
    with pytest.raises(_lib.UrhGpuError):
        estimators.message_ranges_dev(pipe, i8, 0.1, qad_ask=torch.empty(4096, dtype=torch.float32, device="cuda"))


def test_plateau_decisions_on_device_counts_equal_the_sequence_path(pipe):
    """urhgpu_msg_plateau_decisions (plateau lengths counted on the GPU, sequences fetched only for messages with glitches) gives the
    (tolerance, bit length) pairs of urhgpu_msg_plateaus + urhgpu_msg_bit_lengths: clean and glitchy OOK messages, messages without a
    center; and a message with more distinct lengths than the device table holds (decided from its sequence)"""
    import torch
    from urh_amd import estimators
    rng = np.random.default_rng(21)
    n = 3_000_000
    x = np.zeros(n, np.float32)
    ranges, cen = [], []
    pos = 1000
    k = 0
    while pos < n - 400_000:
        kind = k % 4
        nsym = int(rng.integers(300, 3000))
        sps = int(rng.choice([20, 57, 100]))
        bits = rng.integers(0, 2, nsym)
        sym = np.repeat(bits, sps).astype(np.float32)
        if kind == 1:                                      # glitches: single-sample flips -> positive tolerance
            flips = rng.random(len(sym)) < 0.002
            sym = np.where(flips, 1 - sym, sym)
        seg = (0.1 + 0.8 * sym + 0.01 * rng.standard_normal(len(sym))).astype(np.float32)
        x[pos:pos + len(seg)] = seg
        ranges.append((pos, pos + len(seg)))
        cen.append(np.nan if kind == 2 and k % 8 == 2 else 0.5)
        pos += len(seg) + int(rng.integers(100, 5000))
        k += 1
    for a, b in ((pos, pos), (pos + 10, pos + 11), (pos + 20, pos + 25), (pos + 40, pos + 400)):      # degenerate messages: empty, one sample, a few
        x[a:b] = 0.9
        ranges.append((a, b)); cen.append(0.5)
    ranges = np.ascontiguousarray(ranges, dtype=np.int64)
    cen = np.array(cen, dtype=np.float64)
    dev = torch.from_numpy(x).cuda()
    tol, bl = estimators._plateau_decisions(pipe, dev, ranges, cen, 25)
    lens, off = estimators._plateaus_raw(pipe, dev, ranges, cen, 25)
    assert not (off < 0).any() and not (tol == -3).any()
    want = estimators._bit_lengths_raw(lens, off, lambda m: lens[int(off[m]):int(off[m + 1])])
    # (-2: "numpy's order decides" is resolved by _bit_lengths_raw; compare what the native call itself says)
    import ctypes as C
    from urh_amd import _lib
    t2, b2 = np.zeros(len(ranges), np.int64), np.zeros(len(ranges), np.int64)
    _lib.check(_lib.load().urhgpu_msg_bit_lengths(lens.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), len(ranges),
                                                  t2.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(tol, t2) and np.array_equal(bl, b2), (tol.tolist(), t2.tolist(), bl.tolist(), b2.tolist())
    assert (t2 > 0).sum() >= 3 and (t2 == 0).sum() >= 3 and len(want) == len(ranges)
    del dev
    # one message whose first quarter holds some 3500 plateaus of as many distinct lengths (190 .. 3789: no glitches, tolerance 0): more
    # than the device's table takes, so its sequence is fetched -- same answer as the two-step path
    lengths = rng.permutation(np.arange(190, 3790))
    level = (np.arange(len(lengths)) % 2).astype(np.float32)
    head = np.repeat(level, lengths)
    big = np.concatenate([np.zeros(64, np.float32), 0.1 + 0.8 * head, np.full(int(2.9 * len(head)), 0.9, np.float32)])
    r2 = np.array([[64, len(big)]], dtype=np.int64)
    c2 = np.array([0.5])
    dev2 = torch.from_numpy(big).cuda()
    tol_b, bl_b = estimators._plateau_decisions(pipe, dev2, r2, c2, 25)
    lens_b, off_b = estimators._plateaus_raw(pipe, dev2, r2, c2, 25)
    assert off_b[1] >= 3300 and len(np.unique(lens_b[:off_b[1]])) > 3072, off_b
    t3, b3 = np.zeros(1, np.int64), np.zeros(1, np.int64)
    _lib.check(_lib.load().urhgpu_msg_bit_lengths(lens_b.ctypes.data_as(C.c_void_p), off_b.ctypes.data_as(C.c_void_p), 1,
                                                  t3.ctypes.data_as(C.c_void_p), b3.ctypes.data_as(C.c_void_p)))
    assert (int(tol_b[0]), int(bl_b[0])) == (int(t3[0]), int(b3[0])) and t3[0] == 0, (tol_b, bl_b, t3, b3)


def test_estimate_takes_the_numpy_merge_when_the_device_merge_is_borderline(pipe):
    """a pulse length within rounding of mean +- std makes urhgpu_message_ranges_dev hand the OOK merge to numpy: forced here, the
    estimate must come out the same"""
    import torch
    from urh_amd import _lib, estimators
    rng = np.random.default_rng(3)
    n, sps = 600_000, 40
    env = np.zeros(n, np.float32)
    pos = 3000
    while pos < n - 20_000:
        bits = rng.integers(0, 2, 120); bits[0] = 1
        sym = np.repeat(bits, sps).astype(np.float32)
        env[pos:pos + len(sym)] = sym
        pos += len(sym) + int(rng.integers(40, 200)) * sps
    iq = (env * np.exp(2j * np.pi * 0.01 * np.arange(n))).astype(np.complex64) + \
        (0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    dev = torch.from_numpy(np.ascontiguousarray(iq.view(np.float32).reshape(-1, 2))).cuda()
    want = estimators.estimate_dev(pipe, dev, noise=0.1, modulation="OOK")
    lib = _lib.load()
    lib.urhgpu_test_force_merge_ambiguous(1)
    try:
        assert estimators.message_ranges_dev(pipe, dev, 0.1)[4] is True
        got = estimators.estimate_dev(pipe, dev, noise=0.1, modulation="OOK")
    finally:
        lib.urhgpu_test_force_merge_ambiguous(0)
    assert want is not None and got == want, (want, got)


def test_segment_messages_equals_oracle(pipe, oracle):
    import torch
    from urh_amd import estimators
    for n, seed in ((1, 0), (9, 1), (10, 2), (11, 3), (500, 4), (70_001, 5), (200_000, 6), (262_144, 7)):
        iq = _bursty(n, seed)
        for nt in (0.3, 0.0, 5.0):
            want = oracle.segment_messages_from_magnitudes(oracle.get_magnitudes(iq), nt)
            got = estimators.segment_messages_dev(pipe, torch.from_numpy(iq).cuda(), nt)
            assert got == [(int(a), int(b)) for a, b in want], (n, nt, got[:4], want[:4])
            _check_message_ranges(pipe, oracle, iq, nt, want)


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16])
def test_segment_messages_integer_captures(pipe, oracle, dtype):
    import torch
    from urh_amd import estimators
    for n, seed in ((12, 1), (5_000, 2), (120_000, 3)):
        iq = synth_fsk(n, sps=50, seed=seed, noise=0.02, pause_every=max(n // 4, 1), pause_len=n // 9, dtype=dtype)
        if dtype == np.uint16:
            iq[n // 2:n // 2 + 30] = 65535                         # C int overflow -> NaN magnitudes -> "not above"
        mags = oracle.get_magnitudes(iq)
        finite = mags[np.isfinite(mags)]
        for nt in (float(np.median(finite)) if len(finite) else 1000.0, 0.0, 1e9, float("nan")):
            want = oracle.segment_messages_from_magnitudes(mags, nt)
            got = estimators.segment_messages_dev(pipe, torch.from_numpy(iq).cuda(), nt)
            assert got == [(int(a), int(b)) for a, b in want], (np.dtype(dtype).name, n, nt, got[:3], want[:3])
            _check_message_ranges(pipe, oracle, iq, nt, want)


def test_detect_center_equals_numpy(pipe, oracle):
    """the GPU passes reproduce numpy's float32 pairwise np.var and np.histogram exactly -> identical center"""
    import torch
    from urh_amd import estimators
    cases = []
    for name in GOLDEN_CASES:
        cases.append((name, load_golden(name)["qad"]))
    for n, seed in ((100, 1), (4_000, 2), (131_072, 3), (1_000_003, 4)):
        iq = synth_fsk(n, sps=100, seed=seed, noise=0.05, pause_every=max(n // 3, 1), pause_len=n // 20)
        cases.append((f"fsk{n}", oracle.afp_demod(iq, 0.2, "FSK", 2)))
        cases.append((f"ask{n}", oracle.afp_demod(iq, 0.0, "ASK", 2)))
    for name, qad in cases:
        want = oracle.detect_center(qad)
        got = estimators.detect_center_dev(pipe, torch.from_numpy(np.ascontiguousarray(qad, np.float32)).cuda())
        assert (want is None and got is None) or (want is not None and got is not None and float(want) == float(got)), (name, want, got)
        if len(qad) > 5000:
            want = oracle.detect_center(qad, max_size=3000)
            got = estimators.detect_center_dev(pipe, torch.from_numpy(np.ascontiguousarray(qad, np.float32)).cuda(), max_size=3000)
            assert (want is None and got is None) or float(want) == float(got), (name, "max_size", want, got)


def test_plateau_lengths_equal_oracle(pipe, oracle):
    import torch
    from urh_amd import estimators
    for n, seed in ((0, 0), (3, 1), (1000, 2), (60_000, 3), (150_001, 4)):
        iq = synth_fsk(max(n, 4), sps=40, seed=seed, noise=0.1)[:n]
        qad = oracle.afp_demod(iq, 0.0, "FSK", 2) if n else np.zeros(0, np.float32)
        for center in (0.0, 0.05):
            for pct in (25, 100):
                want = oracle.get_plateau_lengths(qad, center, pct)
                got = estimators.get_plateau_lengths_dev(pipe, torch.from_numpy(qad).cuda(), center, pct)
                assert np.array_equal(want, got), (n, center, pct, want[:6], got[:6], len(want), len(got))


def test_batched_message_statistics_equal_oracle(pipe, oracle):
    """urhgpu_msg_center_stats / urhgpu_msg_plateaus (all messages of a capture in one pass each) against the oracle's detect_center
    and get_plateau_lengths message by message: message sizes from 0 to 300 000 samples (partial tiles, partial pairwise chunks),
    noise-gated stretches inside messages, a constant message (zero variance -> no center), an all-noise message."""
    import torch
    from urh_amd import estimators
    rng = np.random.default_rng(404)
    n = 1_500_000
    qad = (np.repeat(rng.integers(0, 2, n // 50 + 1), 50)[:n] * 1.1 - 0.55 + 0.08 * rng.standard_normal(n)).astype(np.float32)
    qad[rng.integers(0, n, 20_000)] = -4.0                         # noise-gated samples sprinkled in
    qad[400_000:400_700] = -4.0
    bounds = [(0, 0), (5, 9), (10, 137), (200, 8392), (9000, 9000 + 8192), (20_000, 20_000 + 16_384 + 77), (50_000, 350_000),
              (360_000, 500_001), (600_000, 600_050), (700_000, 900_123), (1_000_000, 1_000_000 + 4096), (1_100_000, 1_499_999)]
    qad[600_000:600_050] = 0.25                                    # constant: zero variance
    qad[1_000_000:1_000_000 + 4096] = -4.0                         # all noise: nothing kept
    dev = torch.from_numpy(qad).cuda()
    centers = estimators.centers_batched(pipe, dev, bounds)
    for (a, b), c in zip(bounds, centers):
        want = oracle.detect_center(qad[a:b]) if b > a else None
        assert (c is None and want is None) or (c is not None and want is not None and float(c) == float(want)), ((a, b), c, want)
    assert sum(c is not None for c in centers) >= 6
    use = [c if c is not None else (0.1 if i % 2 else None) for i, c in enumerate(centers)]
    plats = estimators.plateau_lengths_batched(pipe, dev, bounds, use)
    for (a, b), c, got in zip(bounds, use, plats):
        want = oracle.get_plateau_lengths(qad[a:b], c, 25) if c is not None else np.zeros(0, np.uint64)
        assert np.array_equal(np.asarray(got, dtype=np.uint64), np.asarray(want, dtype=np.uint64)), ((a, b), c, len(got), len(want))
    # a long plateau beyond the first search window (25 % + 65 536 samples): the batched pass reports it, the caller repeats it
    flat = np.full(600_000, 0.5, np.float32)
    flat[:10_000] = np.where((np.arange(10_000) // 100) % 2 == 0, 0.5, -0.5)
    flat[590_000:] = -0.5
    got = estimators.plateau_lengths_batched(pipe, torch.from_numpy(flat).cuda(), [(0, 600_000)], [0.0])[0]
    assert np.array_equal(np.asarray(got, np.uint64), oracle.get_plateau_lengths(flat, 0.0, 25))


def test_batched_centers_dense_messages_and_tied_peaks(pipe, oracle):
    """messages without noise-gated samples are not compacted (the statistics read the capture itself); a histogram whose second and
    third peak hold the same count is handed to numpy (np.argsort's order of equal keys decides in the reference)"""
    import torch
    from urh_amd import estimators
    rng = np.random.default_rng(7)
    n = 700_000
    qad = (np.repeat(rng.integers(0, 2, n // 40 + 1), 40)[:n] * 0.9 + 0.05 + 0.03 * rng.standard_normal(n)).astype(np.float32)
    qad[300_000:300_100] = -4.0                                      # one gated stretch: that message alone is compacted
    qad[650_000:650_010] = np.nan
    tied = np.zeros(40_000, np.float32)
    tied[50::100] = 10.0
    tied[51::100] = 20.0
    qad[400_000:440_000] = tied
    bounds = [(0, 100_000), (100_003, 250_001), (250_001, 350_000), (400_000, 440_000), (450_000, 450_001), (460_000, 700_000)]
    dev = torch.from_numpy(qad).cuda()
    centers = estimators.centers_batched(pipe, dev, bounds)
    for (a, b), c in zip(bounds, centers):
        want = oracle.detect_center(qad[a:b])
        assert (c is None and want is None) or (c is not None and want is not None and float(c) == float(want)), ((a, b), c, want)
    assert sum(c is not None for c in centers) >= 5
    # the tied message really went through the tie path
    import ctypes as C
    from urh_amd import _lib
    ranges = np.array(bounds, np.int64)
    stats, cen, flag = np.zeros((len(bounds), 8)), np.zeros(len(bounds)), np.zeros(len(bounds), np.int32)
    _lib.check(_lib.load().urhgpu_msg_center_stats(pipe.ctx.handle, C.c_void_p(dev.data_ptr()), n, ranges.ctypes.data_as(C.c_void_p), len(bounds),
                                                   4096, stats.ctypes.data_as(C.c_void_p), None, cen.ctypes.data_as(C.c_void_p),
                                                   flag.ctypes.data_as(C.c_void_p)))
    assert flag.tolist() == [1, 1, 1, 3, 0, 1] and stats[0, 0] == 100_000 and stats[2, 0] == 99_999 - 100
    # a nearly constant message: variance 1e-8 over a range of 4e-4 -> 40 000 bins, more than the pool holds (flag 2): that message
    # goes through the single-message path
    flat = (1.0 + 1e-4 * rng.standard_normal(50_000)).astype(np.float32)
    both = np.concatenate([qad[:100_000], flat])
    got = estimators.centers_batched(pipe, torch.from_numpy(both).cuda(), [(0, 100_000), (100_000, 150_000)])
    want = [oracle.detect_center(both[:100_000]), oracle.detect_center(flat)]
    assert all((g is None and w is None) or float(g) == float(w) for g, w in zip(got, want)), (got, want)
    # ONE native call for centers + plateau decisions (urhgpu_msg_estimate) equals the two calls, the tied message (flag 3) and the
    # nearly constant one (flag 2) settled by the host in between
    for sig, bnd in ((dev, bounds), (torch.from_numpy(both).cuda(), [(0, 100_000), (100_000, 150_000)])):
        r = np.array(bnd, np.int64)
        c1, t1, b1 = estimators.centers_and_decisions(pipe, sig, r, 25)
        c2 = estimators.centers_array(pipe, sig, r)
        t2, b2 = estimators._plateau_decisions(pipe, sig, r, c2.astype(np.float32).astype(np.float64), 25)
        assert np.array_equal(c1, c2, equal_nan=True) and np.array_equal(t1, t2) and np.array_equal(b1, b2), (c1, c2, t1, t2, b1, b2)


def test_branch_free_sincosf_equals_branchy_for_every_float_below_120(pipe):
    """glibc_sincosf.h: urh_sincosf_fast (what the Costas loop evaluates per sample) against urh_sinf / urh_cosf -- which
    tests/test_sincosf_port.py pins to the host libm -- on all 2.25e9 floats with |y| < 120"""
    import ctypes as C
    from urh_amd import _lib
    bad = C.c_uint64(123)
    _lib.check(_lib.load().urhgpu_test_sincosf_fast_dev(pipe.ctx.handle, C.byref(bad)))
    assert bad.value == 0


def test_detect_modulation_on_device(pipe):
    """urhgpu_detect_modulation_dev against detect_modulation in numpy (and the real reference's, where staged): same label for every
    message, variances within 1e-4 relative (double-precision radix-2 FFTs here, numpy's single-precision forward transforms there):
    synthetic OOK / ASK / FSK / PSK / noise messages on both transform paths (<= 2048 points in LDS, longer through HBM), messages with
    zeros, tiny and empty ones; the messages of the golden captures."""
    import torch
    import ref_python
    import numpy_estimators
    from urh_amd import estimators
    ref_dm = None
    if ref_python.available():
        ref_python.setup()
        from urh.ainterpretation import AutoInterpretation as AI
        ref_dm = AI.detect_modulation
    rng = np.random.default_rng(2024)
    msgs = []

    def tone(n, kind):
        t = np.arange(n)
        sym = np.repeat(rng.integers(0, 2, n // 40 + 1), 40)[:n]
        if kind == "FSK":
            x = np.exp(1j * np.cumsum(np.where(sym == 1, 0.25, -0.25)))
        elif kind == "ASK":
            x = (0.3 + 0.7 * sym) * np.exp(1j * 0.2 * t)
        elif kind == "PSK":
            x = np.exp(1j * (0.2 * t + np.pi * sym))
        elif kind == "OOK1":
            x = np.exp(1j * 0.2 * t)                       # one unmodulated pulse
        else:
            x = np.zeros(n, complex)
        return (x + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    for kind in ("FSK", "ASK", "PSK", "OOK1", "noise"):
        for n in (300, 1500, 2048, 2049, 5000, 70_000):
            msgs.append(tone(n, kind))
    z = tone(4000, "FSK"); z[100:110] = 0; msgs.append(z)           # more than 3 zeros: OOK
    z = tone(4000, "FSK"); z[7] = 0; msgs.append(z)                 # up to 3 zeros are only dropped
    msgs += [tone(10, "FSK"), tone(17, "ASK"), np.zeros(50, np.complex64), np.zeros(0, np.complex64), tone(33, "PSK")]
    capture = np.concatenate(msgs)
    bounds, o = [], 0
    for mm in msgs:
        bounds.append((o, o + len(mm))); o += len(mm)
    dev = torch.from_numpy(capture.view(np.float32).reshape(-1, 2)).cuda()
    labels, variances = estimators.detect_modulation_dev(pipe, dev, bounds, return_variances=True)
    seen = set()
    for k, (mm, lab) in enumerate(zip(msgs, labels)):
        want = numpy_estimators.detect_modulation(mm.copy())
        assert lab == want, (k, len(mm), lab, want, variances[k])
        if ref_dm is not None:
            assert lab == ref_dm(mm.copy()), (k, len(mm))
        seen.add(lab)
    assert {"FSK", "ASK", "PSK", "OOK", None} <= seen, seen
    # variances against numpy on one long and one short message
    for k in (5, 1):
        d = msgs[k][np.abs(msgs[k]) > 0]
        d = d / np.abs(np.max(d))
        w1 = np.abs(numpy_estimators.cwt_haar(d, scale=4)); w2 = np.abs(numpy_estimators.cwt_haar(d / np.abs(d), scale=4))
        want = [np.var(w1), np.var(w2), np.var(numpy_estimators.median_filter(w1, 11)), np.var(numpy_estimators.median_filter(w2, 11))]
        assert np.allclose(variances[k], want, rtol=1e-4, atol=1e-9), (k, variances[k], want)
    # the golden captures: every message, device label == numpy label
    from urh_amd.pipeline import DevicePipeline
    for name in GOLDEN_CASES:
        g = load_golden(name)
        iq = g["iq"]
        if iq.dtype != np.float32:
            continue
        d = torch.from_numpy(iq).cuda()
        segs = estimators.segment_messages_dev(pipe, d, g["noise_threshold"])[:100]
        if not segs:
            continue
        got = estimators.detect_modulation_dev(pipe, d, segs)
        cplx = iq.view(np.complex64).reshape(-1)
        assert got == [numpy_estimators.detect_modulation(cplx[a:b].copy()) for a, b in segs], name


def test_estimate_equals_reference_goldens(pipe):
    """AutoInterpretation.estimate on the GPU vs what the real reference returned for the same captures
    (tests/golden/estimates.json, made by tests/golden/make_estimate_golden.py): identical dict, floats included."""
    import json
    import os
    import torch
    from conftest import GOLDEN_DIR
    from urh_amd import estimators
    want = json.load(open(os.path.join(GOLDEN_DIR, "estimates.json")))
    for key, w in want.items():
        name, mod, how = key.split("|")
        if mod == "PSK":
            # the reference's Costas demodulator never writes result[0] (np.empty, signal_functions.pyx:265, :289): what
            # estimate() sees there is uninitialised memory, so its PSK estimates are not reproducible to the last digit
            continue
        g = load_golden(name)
        noise = None if how == "auto" else g["noise_threshold"]
        dev_iq = torch.from_numpy(g["iq"]).cuda()
        if mod == "detect":                          # modulation=None: detect_modulation_for_messages decides
            mod = None
            nz = estimators.detect_noise_level_dev(pipe, dev_iq) if noise is None else noise
            if estimators.detect_modulation_for_messages_dev(dev_iq, estimators.segment_messages_dev(pipe, dev_iq, nz)) == "PSK":
                continue                             # same caveat as above
        got = estimators.estimate_dev(pipe, dev_iq, noise=noise, modulation=mod)
        if w is None:
            assert got is None, (key, got)
            continue
        assert got is not None, key
        assert got["modulation_type"] == w["modulation_type"] and int(got["bit_length"]) == w["bit_length"] \
            and int(got["tolerance"]) == w["tolerance"], (key, got, w)
        assert float(got["center"]) == w["center"] and float(got["noise"]) == w["noise"], (key, got, w)


# ---- BASELINE.json full-size configurations: size-independent properties -------------------------------------------------
def test_full_size_fsk_1gib_properties(pipe):
    """configs[1] at full size (1 GiB complex64 2-FSK @ 100 samples/symbol): the recovered bits equal the transmitted
    bits (one message: the 76-sample gaps between segments are short pauses), the pulse-table lengths add up to the
    capture, and the 8-way sharded pass (8 simulated ranks, 128 MiB each: configs[3] per-GPU logic) gives the same
    rows / bits / positions / qad bit for bit."""
    import torch
    from test_sharding import run_threads
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import shard_bounds, stitch
    from urh_amd.synth import fsk_capture
    dev = torch.device("cuda", 0)
    segs, sps, tol = 128, 100, 5
    iq, tx = fsk_capture(segs, dev, seed=1234, sps=sps)
    n = iq.shape[0]
    assert n == 1 << 27
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, tol, sps, 0.1, 8, True)
    res = pipe.iq_to_bits(iq, p, want_qad=True)
    res.check_capacity()
    rows = res.ppseq()
    bits, off, pauses, pos, poff = res.flat()
    # lengths: first row r+1, last row n-1-r_last-tol => sum = n - tol
    assert int(rows[:, 1].sum()) == n - tol
    assert len(pauses) == 1 and off[-1] == len(bits)
    # transmitted symbols: per segment 10485 symbols, then a 76-sample gap that demodulates to noise (random states)
    nsym = tx.shape[1]
    txb = tx.cpu().numpy()
    got = bits
    # align per segment through bit_sample_pos: bit k of the message starts at sample pos[k]
    seg_of = pos[:len(bits)] // (1 << 20)
    sym_of = (pos[:len(bits)] % (1 << 20) + sps // 2) // sps
    ok = sym_of < nsym
    errors = int((got[ok] != txb[seg_of[ok], sym_of[ok]]).sum())
    assert ok.sum() > 0.999 * segs * nsym and errors < 1e-4 * ok.sum(), (int(ok.sum()), errors)
    want = (rows, bits, off, pauses, pos, poff)
    qad_sum = int(res.qad.view(torch.int32).to(torch.int64).sum().item())
    world = 8
    bounds = shard_bounds(n, world)
    shards = [iq[a:b] for a, b in bounds]
    out = run_threads(world, lambda r: GpuShardEngine(0), shards, bounds, n, p)
    got_all = stitch(out)
    for k, (a, b) in enumerate(zip(got_all, want)):
        assert np.array_equal(a, b), (k, len(a), len(b))
    assert sum(int(r.qad.view(torch.int32).to(torch.int64).sum().item()) for r in out) == qad_sum


def test_spec_capture_bytes_equal_reference_generator(pipe, oracle):
    """urh_amd.synth.spec_fsk_capture builds SURVEY §8(d) config 2's capture byte for byte: two segments through the GPU generator
    equal the same segments through the oracle's modulate_c (and the real reference's, where oracle/_ref is built) + numpy AWGN."""
    import array
    import build_ref
    from urh_amd.synth import spec_fsk_bits, spec_fsk_capture
    iq, bits = spec_fsk_capture(2, "cuda:0", first_segment=5)
    got = iq.cpu().numpy()
    gens = [oracle.modulate_c]
    if build_ref.built():
        gens.append(build_ref.import_ref()[0].modulate_c)
    for gen in gens:
        for j, k in enumerate((5, 6)):
            b = np.random.default_rng(1234 + k).integers(0, 2, 10485)
            assert np.array_equal(b, bits[j]) and np.array_equal(b, spec_fsk_bits(k))
            seg = np.asarray(gen(array.array("B", b.tolist()), 100, "FSK", array.array("f", [-20e3, 20e3]), 1, 1.0, 40e3, 0.0, 1e6, 76, 0),
                             dtype=np.float32)
            assert seg.shape == (1 << 20, 2)
            want = seg + np.float32(0.05) * np.random.default_rng(5678 + k).standard_normal((1 << 20, 2)).astype(np.float32)
            assert want.dtype == np.float32
            assert np.array_equal(got[j << 20:(j + 1) << 20].view(np.uint32), want.view(np.uint32)), (gen, k)


def full_size_reference(iq_host, p, oracle):
    """qad / pulse table / flat bits of the CPU side for a full-size capture: the REAL reference's Cython functions (oracle/_ref)
    when they are built -- afp_demod is an OpenMP prange --, else the C restatement; the tail (pure Python in the reference, a
    minute at this size) is the C restatement, which tests/test_oracle.py pins against the reference's Python."""
    import build_ref
    order = 2 ** p.bits_per_symbol
    if build_ref.built():
        sfr = build_ref.import_ref()[0]
        qad = np.asarray(sfr.afp_demod(iq_host, p.noise_threshold, p.modulation_type, order, p.costas_loop_bandwidth))
        pp = np.asarray(sfr.grab_pulse_lens(qad, p.center, p.tolerance, p.modulation_type, p.samples_per_symbol, p.bits_per_symbol,
                                            p.center_spacing))
    else:
        qad = oracle.afp_demod(iq_host, p.noise_threshold, p.modulation_type, order, p.costas_loop_bandwidth)
        pp = oracle.grab_pulse_lens(qad, p.center, p.tolerance, p.modulation_type, p.samples_per_symbol, p.bits_per_symbol, p.center_spacing)
    flat = oracle.ppseq_to_bits_flat(pp, p.samples_per_symbol, p.bits_per_symbol, True, p.pause_threshold)
    return qad, pp, flat


@pytest.mark.parametrize("variant", ["2", "2b"])
def test_full_size_fsk_1gib_bit_exact(pipe, oracle, variant):
    """BASELINE.json configs[1] at FULL size on the bytes SURVEY §8(d) config 2 specifies (variant 2b: bursty, 2 076-sample gaps,
    noise threshold 0.2, 128 messages): the float32 demodulated signal (uint32 view), the pulse table, the bits, pauses,
    message offsets and bit_sample_pos of the 2^27-sample capture equal the reference's, element for element."""
    import torch
    from urh_amd.pipeline import DemodParams
    from urh_amd.synth import spec_fsk_capture
    segs = 128
    if variant == "2":
        iq, tx = spec_fsk_capture(segs, "cuda:0")
        p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    else:
        iq, tx = spec_fsk_capture(segs, "cuda:0", seg_len=1 << 20, sps=100, n_symbols=10465)
        p = DemodParams("FSK", 1, 0.2, 0.0, 1.0, 5, 100, 0.1, 8, True)
    n = iq.shape[0]
    assert n == 1 << 27
    res = pipe.iq_to_bits_checked(iq, p, want_qad=True)
    rows = res.ppseq()
    got_flat = res.flat()
    got_qad = res.qad.cpu().numpy()
    host = iq.cpu().numpy()
    del iq
    qad, pp, flat = full_size_reference(host, p, oracle)
    assert int((got_qad.view(np.uint32) != qad.view(np.uint32)).sum()) == 0
    assert np.array_equal(rows, pp), (len(rows), len(pp))
    for k, (a, b) in enumerate(zip(got_flat, flat)):
        assert np.array_equal(a, b), (k, len(a), len(b))
    n_msg = len(flat[2])
    assert n_msg == (1 if variant == "2" else 128), n_msg
    # sanity vs the transmitter: bit k of a message starts at sample pos[k]
    bits, off, pauses, pos, poff = got_flat
    errors = total = 0
    for m in range(n_msg):
        bp = pos[poff[m]:poff[m] + (off[m + 1] - off[m])]
        seg_of, sym_of = bp // (1 << 20), (bp % (1 << 20) + 50) // 100
        ok = sym_of < tx.shape[1]
        errors += int((bits[off[m]:off[m + 1]][ok] != tx[seg_of[ok], sym_of[ok]]).sum())
        total += int(ok.sum())
    assert total > 0.99 * tx.size and errors < 1e-3 * total, (total, errors)


def test_config3_ook_fir_auto_noise_pipeline(pipe, sf, oracle):
    """configs[2] (OOK + 64-tap FIR band-pass + automatic noise threshold) at a size the oracle finishes in seconds:
    Signal.filter_range semantics (fir_filter), detect_noise_level, ASK demodulation and digitization, each stage and
    the final bits identical to the oracle."""
    import torch
    from urh_amd import estimators
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(4321)
    n, sps = 1 << 20, 100
    chips = np.repeat(np.array([[1, 0], [0, 1]])[rng.integers(0, 2, n // sps // 2 + 1)].reshape(-1), sps)[:n]   # Manchester
    env = chips.astype(np.float64)
    env[:n // 8] = 0                                                        # leading noise-only stretch (for the estimator)
    env[n // 2:n // 2 + 60_000] = 0                                         # a long pause: two messages
    ph = 2 * np.pi * 0.04 * np.arange(n)
    iq = (env[:, None] * np.stack([np.cos(ph), np.sin(ph)], 1) + 0.02 * rng.standard_normal((n, 2))).astype(np.float32)
    m = 64
    k = np.arange(m) - (m - 1) / 2
    taps = (np.sinc(2 * 0.02 * k) * 2 * 0.02 * np.blackman(m) * np.exp(2j * np.pi * 0.04 * k)).astype(np.complex64)
    x = iq.view(np.complex64).reshape(-1)
    filt_want = oracle.fir_filter(x, taps)
    filt = sf.fir_filter(x, taps)
    assert cbits_equal(filt, filt_want)
    fiq = np.ascontiguousarray(filt.view(np.float32).reshape(-1, 2))
    noise_want = oracle.detect_noise_level(oracle.get_magnitudes(fiq))
    d_fiq = torch.from_numpy(fiq).cuda()
    noise = estimators.detect_noise_level_dev(pipe, d_fiq)
    assert noise == noise_want and noise > 0
    qad_want = oracle.afp_demod(fiq, noise, "ASK", 2)
    center = float(oracle.detect_center(qad_want))
    p = DemodParams("ASK", 1, noise, center, 1.0, 5, sps, 0.1, 8, True)
    res = pipe.iq_to_bits(d_fiq, p, want_qad=True)
    assert bits_equal(res.qad.cpu().numpy(), qad_want)
    assert float(estimators.detect_center_dev(pipe, res.qad)) == center
    pp = oracle.grab_pulse_lens(qad_want, center, 5, "ASK", sps, 1, 1.0)
    assert np.array_equal(res.ppseq(), pp)
    fb = oracle.ppseq_to_bits_flat(pp, sps, 1, True, 8)
    assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))
    assert len(fb[2]) >= 2                                                 # at least two messages


@pytest.mark.parametrize("order", [2, 4])
def test_costas_parallel_chain_large(sf, oracle, order):
    """configs[4]-like PSK capture of 4 Mi samples (carrier 0.04 cycles/sample, AWGN, a long gated pause, an un-gated
    noise-only stretch): the speculative chunk evaluation of the Costas loop is bit-exact, and most chunks resolve
    through a matching candidate."""
    from urh_amd import _lib
    rng = np.random.default_rng(100 + order)
    n, sps = 1 << 22, 100
    sym = rng.integers(0, order, n // sps + 1)
    phases = (np.array([-135, -45, 45, 135]) if order == 4 else np.array([-90, 90]))[sym] * np.pi / 180
    ph = np.repeat(phases, sps)[:n] + 2 * np.pi * 0.04 * np.arange(n)
    iq = np.stack([np.cos(ph), np.sin(ph)], 1) + 0.1 * np.sqrt(0.5) * rng.standard_normal((n, 2))
    iq[1_000_000:1_300_000] *= 0.01                          # gated pause (below the noise threshold)
    iq[2_000_000:2_050_000] = 0.5 * rng.standard_normal((50_000, 2))   # un-gated noise: the loop wanders
    iq = iq.astype(np.float32)
    want = oracle.afp_demod(iq, 0.2, "PSK", order, 0.1)
    got = sf.afp_demod(iq, 0.2, "PSK", order, 0.1)
    assert bits_equal(got[1:], want[1:]), int((got[1:] != want[1:]).sum())
    stats = _lib.default_context().costas_stats()
    assert sum(stats[:3]) == (n - 1 + 4095) // 4096 - 1
    # the 300k-sample gated pause is 73 chunks that no candidate can match; everything else should
    assert stats[0] > 0.85 * sum(stats[:3]), stats


@pytest.mark.parametrize("bandwidth", [0.05, 0.1, 0.25])
def test_costas_warm_up_follows_loop_bandwidth(sf, oracle, bandwidth):
    """the candidates' warm-up length is derived from the loop bandwidth (512 samples at the default 0.1): still bit-exact, and on a
    noisy capture (SNR 10 dB) with a carrier offset practically every chunk is carried by a candidate"""
    from urh_amd import _lib
    for order, offset in ((4, 0.013), (2, -0.006)):
        rng = np.random.default_rng(int(bandwidth * 1000) + order)
        n, sps = 1 << 20, 64
        sym = rng.integers(0, order, n // sps + 1)
        phases = (np.array([-135, -45, 45, 135]) if order == 4 else np.array([-90, 90]))[sym] * np.pi / 180
        ph = np.repeat(phases, sps)[:n] + 2 * np.pi * offset * np.arange(n)
        iq = (np.stack([np.cos(ph), np.sin(ph)], 1) + 0.316 * np.sqrt(0.5) * rng.standard_normal((n, 2))).astype(np.float32)
        want = oracle.afp_demod(iq, 0.05, "PSK", order, bandwidth)
        got = sf.afp_demod(iq, 0.05, "PSK", order, bandwidth)
        assert bits_equal(got[1:], want[1:]), (order, int((got[1:] != want[1:]).sum()))
        stats = _lib.default_context().costas_stats()
        assert stats[0] >= 0.97 * sum(stats[:3]), (bandwidth, order, stats)


def test_fir_with_fused_noise_statistics(pipe, oracle):
    """urhgpu_fir_filter_stats_dev: the filtered samples are those of urhgpu_fir_filter_dev bit for bit, and the noise threshold from
    the fused chunk statistics equals detect_noise_level on the filtered signal (oracle), for sizes whose 1 % chunks cut tiles anywhere,
    a capture with a left halo, and one too small for the fused path (chunk < one tile)."""
    import torch
    from urh_amd import estimators
    from urh_amd.synth import spec_fir_taps
    taps = spec_fir_taps()
    rng = np.random.default_rng(31)
    for n in (1_000_003, 409_600, 204_799, 2_500_000, 150_000):
        env = np.repeat(rng.integers(0, 2, n // 5000 + 1), 5000)[:n]
        env[: n // 10] = 0                                      # a noise-only stretch so that the estimator does not return 0
        x = ((0.9 * env + 0.02 * rng.standard_normal(n)) * np.exp(2j * np.pi * 0.04 * np.arange(n)) + 0.02j * rng.standard_normal(n)).astype(np.complex64)
        dev = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
        filt, noise = estimators.fir_filter_detect_noise_dev(pipe, dev, taps)
        want_f = oracle.fir_filter(x, taps)
        assert np.array_equal(filt.cpu().numpy().reshape(-1).view(np.uint32), want_f.view(np.uint32)), n
        want_noise = oracle.detect_noise_level(oracle.get_magnitudes(want_f.view(np.float32).reshape(-1, 2)))
        assert float(noise) == float(want_noise), (n, noise, want_noise)
        assert n < 2_000_000 or noise > 0
    # tiles with a huge sample go through the checked kernel (k_fir_fast hands them back), statistics epilogue included
    n = 500_000
    x = ((0.5 + 0.02 * rng.standard_normal(n)) * np.exp(2j * np.pi * 0.03 * np.arange(n))).astype(np.complex64)
    x[: n // 10] *= 0.01
    x[123_456] = np.float32(3e24)
    x[400_000] = np.complex64(-2e22j)
    dev = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).cuda()
    filt, noise = estimators.fir_filter_detect_noise_dev(pipe, dev, taps)
    want_f = oracle.fir_filter(x, taps)
    assert np.array_equal(filt.cpu().numpy().reshape(-1).view(np.uint32), want_f.view(np.uint32))
    assert float(noise) == float(oracle.detect_noise_level(oracle.get_magnitudes(want_f.view(np.float32).reshape(-1, 2))))


def test_sharded_fir_halo_then_bits(pipe, oracle):
    """configs[3]-style: 4 simulated ranks, FIR with the left neighbour's 63-sample tail as history, then the sharded
    IQ->bits pass on the filtered shards == one-pass oracle FIR + single-GPU pass."""
    import threading
    import torch
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import ShardedPipeline, ThreadComm, shard_bounds, stitch
    rng = np.random.default_rng(9)
    n, m, world = 400_000, 64, 4
    iq = synth_fsk(n, sps=100, seed=77, noise=0.05, pause_every=n // 3, pause_len=n // 25)
    taps = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64) * np.float32(0.05)
    taps[m // 2] += 1
    want_f = oracle.fir_filter(iq.view(np.complex64).reshape(-1), taps)
    fiq = np.ascontiguousarray(want_f.view(np.float32).reshape(-1, 2))
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    single = pipe.iq_to_bits(torch.from_numpy(fiq).cuda(), p, want_qad=True)
    want = (single.ppseq(),) + tuple(single.flat())
    bounds = shard_bounds(n, world)
    dev_iq, dev_taps = torch.from_numpy(iq).cuda(), torch.from_numpy(taps).cuda()
    shared = ThreadComm.Shared(world)
    out, filt, err = [None] * world, [None] * world, []

    def work(r):
        try:
            sp = ShardedPipeline(GpuShardEngine(0), ThreadComm(shared, r))
            a, b = bounds[r]
            f = sp.fir_filter(dev_iq[a:b], dev_taps)
            filt[r] = f.cpu().numpy()
            out[r] = sp.iq_to_bits(f, p, want_qad=True, pos_base=a, n_total=n)
        except BaseException as e:          # noqa: BLE001
            err.append(e)
            shared.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if err:
        raise err[0]
    assert cbits_equal(np.concatenate(filt).view(np.complex64).reshape(-1), want_f)
    for k, (a, b) in enumerate(zip(stitch(out), want)):
        assert np.array_equal(a, b), k
    # the same with NO halo exchange (round 5): every rank but the first is handed the m + 1 raw samples before its shard with the shard;
    # their last m - 1 are the filter's history, filtering them gives the two filtered samples the demodulation needs
    shared2 = ThreadComm.Shared(world)
    out2, filt2, err2 = [None] * world, [None] * world, []

    def work2(r):
        try:
            sp = ShardedPipeline(GpuShardEngine(0), ThreadComm(shared2, r))
            a, b = bounds[r]
            raw = dev_iq[a - (m + 1):a].contiguous() if r > 0 else None
            f, halo = sp.fir_filter(dev_iq[a:b], dev_taps, left_raw=raw, want_halo=True)
            filt2[r] = f.cpu().numpy()
            out2[r] = sp.iq_to_bits(f, p, want_qad=True, pos_base=a, n_total=n, halo_given=True, left_halo=halo)
        except BaseException as e:          # noqa: BLE001
            err2.append(e)
            shared2.barrier.abort()
    ts = [threading.Thread(target=work2, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if err2:
        raise err2[0]
    assert cbits_equal(np.concatenate(filt2).view(np.complex64).reshape(-1), want_f)
    for k, (a, b) in enumerate(zip(stitch(out2), want)):
        assert np.array_equal(a, b), k


def test_get_protocol_from_signal_goldens(pipe):
    """ProtocolAnalyzer.get_protocol_from_signal (bits, pause, RSSI, timestamp, bit_sample_pos per message, ASK padding with
    message_length_divisor) vs what the real reference put into its Message objects (tests/golden/messages.json)."""
    import json
    import os
    import torch
    from conftest import GOLDEN_DIR
    from urh_amd.pipeline import DemodParams
    from urh_amd.protocol import get_protocol_from_signal_dev
    want = json.load(open(os.path.join(GOLDEN_DIR, "messages.json")))
    for key, msgs in want.items():
        name, divisor = key.split("|")
        g = load_golden(name)
        p = DemodParams(g["modulation_type"], g["bits_per_symbol"], g["noise_threshold"], g["center"], g["center_spacing"],
                        g["tolerance"], g["samples_per_symbol"], g["costas_loop_bandwidth"], g["pause_threshold"], True)
        got = get_protocol_from_signal_dev(pipe, torch.from_numpy(g["iq"]).cuda(), p, message_length_divisor=int(divisor))
        assert len(got) == len(msgs), key
        for a, b in zip(got, msgs):
            assert a.plain_bits_str == b["bits"] and a.pause == b["pause"] and list(a.bit_sample_pos) == b["pos"], key
            assert a.rssi == b["rssi"], (key, a.rssi, b["rssi"])
            if b["pos"][0] != 0:            # a message at sample 0 has timestamp 0, which urh's Message replaces by time.time()
                assert a.timestamp == b["timestamp"], key


@pytest.mark.parametrize("mod", ["FSK", "ASK"])
def test_sharded_pipelined_passes(pipe, mod):
    """The sharded path in pipelined mode (bench.py uses it for --gpus N > 1: the exchange-laden tail of a pass overlaps the hot
    kernel of the next one): four back-to-back passes per rank over two alternating captures, two simulated ranks with
    persistent pipelined engines, the halo exchanged (even passes) or handed over with the shard (odd passes); every pass equals the
    single-GPU result."""
    import threading
    import torch
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import ShardedPipeline, ThreadComm, shard_bounds, stitch
    world = 2
    caps = []
    for seed, n in ((5, 700_001), (6, 524_288)):
        iq = synth_fsk(n, sps=50, seed=seed, noise=0.05, pause_every=n // 4, pause_len=n // 31)
        if mod == "ASK":
            env = np.repeat(np.random.default_rng(seed).integers(0, 2, n // 50 + 1), 50)[:n]
            iq = (iq * (0.05 + 0.95 * env)[:, None]).astype(np.float32)
        p = DemodParams(mod, 1, 0.2, 0.0 if mod == "FSK" else 0.35, 1.0, 3, 50, 0.1, 8, True)
        dev = torch.from_numpy(iq).cuda()
        single = pipe.iq_to_bits(dev, p, want_qad=True)
        want = (single.ppseq().copy(),) + tuple(x.copy() for x in single.flat())
        caps.append((dev, p, want, single.qad.cpu().numpy().copy(), shard_bounds(n, world), n))
    shared = ThreadComm.Shared(world)
    n_it = 4
    results, err = [[None] * world for _ in range(n_it)], []
    host_views = [[None] * world for _ in range(n_it)]

    def work(r):
        try:
            # every pass also delivers its compact blob to pinned host memory (host_results; an absorbed first row of an ASK piece: state -128)
            sp = ShardedPipeline(GpuShardEngine(0, pipelined=True, host_results=True), ThreadComm(shared, r))
            handles = []
            for it in range(n_it):
                dev, p, _, _, bounds, n = caps[(it // 2) % 2] if it >= 2 else caps[it % 2]
                a, b = bounds[r]
                given = it % 2 == 1
                res = sp.iq_to_bits(dev[a:b], p, want_qad=True, pos_base=a, n_total=n, halo_given=given,
                                    left_halo=dev[a - 2:a].clone() if given and r > 0 else None)
                handles.append(res)
                if it >= 2:                                 # looked at two passes later, as a streaming caller would
                    h = handles[it - 2].host().check()
                    host_views[it - 2][r] = (h.ppseq().copy(), h.bits().copy(), h.pauses.copy(), h.bit_sample_pos().copy())
                results[it][r] = (res.piece(), res.qad.cpu().numpy().copy())
            if True:
                for it in (n_it - 2, n_it - 1):
                    h = handles[it].host().check()
                    host_views[it][r] = (h.ppseq().copy(), h.bits().copy(), h.pauses.copy(), h.bit_sample_pos().copy())
        except BaseException as e:          # noqa: BLE001 -- re-raised in the main thread
            err.append(e)
            shared.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        raise err[0]
    for it in range(n_it):
        _, _, want, want_qad, _, _ = caps[(it // 2) % 2] if it >= 2 else caps[it % 2]
        got = stitch([results[it][r][0] for r in range(world)])
        for k, (x, y) in enumerate(zip(got, want)):
            assert np.array_equal(x, y), (mod, it, k, len(x), len(y))
        assert bits_equal(np.concatenate([results[it][r][1] for r in range(world)]), want_qad), (mod, it)
        if True:                                             # the host blobs hold the ranks' pieces: rows, bits, pauses, positions
            for r in range(world):
                pc, hv = results[it][r][0], host_views[it][r]
                assert np.array_equal(hv[0], pc["rows"]) and np.array_equal(hv[1], pc["bits"]) and np.array_equal(hv[2], pc["pauses"]), (it, r)
                assert np.array_equal(hv[3], pc["pos"]), (it, r)


def test_pinned_host_copy_equals_pageable(pipe):
    """BitsResult.to_host_pinned (asynchronous copies into pinned buffers, one synchronisation) returns what ppseq() / flat() return"""
    import torch
    from urh_amd.pipeline import DemodParams
    iq = synth_fsk(700_000, sps=100, seed=5, noise=0.05, pause_every=200_000, pause_len=9000)
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    pool = {}
    for _ in range(2):
        res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True)
        got = res.to_host_pinned(pool)
        want = (res.ppseq(),) + res.flat()
        assert len(got) == 6 and all(np.array_equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize("n", [1 << 22, (1 << 22) + 777])
def test_pipelined_passes_do_not_disturb_each_other(oracle, n):
    """pipelined mode: a burst of back-to-back passes over alternating captures without any synchronisation in between;
    the last two results (kept in separate output slots) are bit-exact, i.e. the hot kernel of pass i+1 did not disturb the
    tail of pass i (alternating scratch) and the tails ran in order.  Whole tiles: the tail stream waits for the hot dispatch's own
    completion signal; with a partial tile at the end (a second, one-workgroup launch) for an event recorded behind both."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    pp = DevicePipeline(0, pipelined=True)
    caps, wants = [], []
    for seed, mod in ((1, "FSK"), (2, "ASK")):
        iq = synth_fsk(n, sps=100, seed=seed, noise=0.05, pause_every=n // 3, pause_len=n // 17)
        if mod == "ASK":
            env = np.repeat(np.random.default_rng(seed).integers(0, 2, n // 100 + 1), 100)[:n]
            iq = (iq * (0.05 + 0.95 * env)[:, None]).astype(np.float32)
        p = DemodParams(mod, 1, 0.1, 0.0 if mod == "FSK" else 0.35, 1.0, 5, 100, 0.1, 8, True)
        qad = oracle.afp_demod(iq, 0.1, mod, 2)
        ppseq = oracle.grab_pulse_lens(qad, p.center, 5, mod, 100, 1, 1.0)
        wants.append((qad, ppseq, oracle.ppseq_to_bits_flat(ppseq, 100, 1, True, 8)))
        caps.append((torch.from_numpy(iq).cuda(), p))
    results = [None, None]
    for i in range(12):
        iq, p = caps[i % 2]
        results[i % 2] = pp.iq_to_bits(iq, p, want_qad=True, slot=i % 2)
    for k in (0, 1):
        qad, ppseq, flat = wants[k]
        res = results[k]
        assert np.array_equal(res.ppseq(), ppseq), k
        assert all(np.array_equal(a, b) for a, b in zip(flat, res.flat())), k
        assert bits_equal(res.qad.cpu().numpy(), qad), k
    pp.ctx.set_pipelined(False)


@pytest.mark.parametrize("case", ["rows", "groups", "ask"])
def test_scans_loop_over_more_tiles_than_their_grid(pipe, oracle, case):
    """The pulse-table scans run on a bounded grid (scan.hpp scan_grid: 512 workgroups of 2048 elements) and loop beyond it:
    a capture with more than 2^20 pulse-table rows ("rows": 4 samples per symbol) and one with more than 2^20 messages
    ("groups": six-sample bursts, four-sample gaps, one sample per symbol, pause threshold 2) -- rows, bits, pauses and bit
    positions identical to the oracle."""
    import torch
    from urh_amd.pipeline import DemodParams
    n = 3 << 22
    if case == "rows":
        iq = synth_fsk(n, sps=4, seed=11, noise=0.02, deviation_hz=200e3)
        p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1, 4, 0.1, 8, True)
    elif case == "ask":                          # the ASK row merge (scan over the un-merged table) beyond its grid as well
        iq = synth_fsk(n, sps=4, seed=13, noise=0.01)
        env = np.repeat(np.random.default_rng(13).integers(0, 2, n // 4 + 1), 4)[:n]
        iq = (iq * (0.2 + 0.8 * env)[:, None]).astype(np.float32)
        p = DemodParams("ASK", 1, 0.0, 0.42, 1.0, 1, 4, 0.1, 8, True)
    else:
        iq = synth_fsk(n, sps=1, seed=12, noise=0.01, deviation_hz=200e3)
        iq[(np.arange(n) % 10) >= 6] *= np.float32(0.01)
        p = DemodParams("FSK", 1, 0.3, 0.0, 1.0, 1, 1, 0.1, 2, True)
    mod = p.modulation_type
    qad = oracle.afp_demod(iq, p.noise_threshold, mod, 2)
    ppseq = oracle.grab_pulse_lens(qad, p.center, 1, mod, p.samples_per_symbol, 1, 1.0)
    flat = oracle.ppseq_to_bits_flat(ppseq, p.samples_per_symbol, 1, True, p.pause_threshold)
    assert len(ppseq) > (1 << 20) and (case != "groups" or len(flat[2]) > (1 << 20))
    res = pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=True)
    res.check_capacity()
    assert np.array_equal(res.ppseq(), ppseq)
    assert all(np.array_equal(a, b) for a, b in zip(flat, res.flat()))
    assert bits_equal(res.qad.cpu().numpy(), qad)


def test_error_codes_on_device(pipe, sf):
    """argument / capacity / unsupported-parameter errors surface as the documented status codes and exceptions"""
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams
    lib, h = _lib.load(), pipe.ctx.handle
    iq = torch.from_numpy(synth_fsk(50_000, sps=20, seed=1, noise=0.05)).cuda()
    # pulse table capacity too small: the device path clamps and reports through counts; the host API returns ERR_CAPACITY + size
    qad = sf.afp_demod(iq.cpu().numpy(), 0.0, "FSK", 2)
    rows = np.zeros((4, 2), np.int64)
    n_rows = C.c_int64(0)
    st = lib.urhgpu_grab_pulse_lens(h, qad.ctypes.data_as(C.c_void_p), len(qad), 0.0, 5, _lib.MOD_FSK, 20, 1, 0.1, 0.0,
                                    rows.ctypes.data_as(C.c_void_p), 4, C.byref(n_rows))
    assert st == _lib.ERR_CAPACITY and n_rows.value == len(sf.grab_pulse_lens(qad, 0.0, 5, "FSK", 20))
    res = pipe.iq_to_bits(iq, DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 20), cap_rows=16)
    with pytest.raises(_lib.UrhGpuError):
        res.check_capacity()
    with pytest.raises(_lib.UrhGpuError):                       # bits_per_symbol > 7
        sf.grab_pulse_lens(qad, 0.0, 5, "FSK", 20, 8)
    with pytest.raises(ValueError):                             # (N, 3) is not an IQ array
        sf.afp_demod(np.zeros((10, 3), np.float32), 0.0, "FSK", 2)
    with pytest.raises(OverflowError):
        sf.grab_pulse_lens(qad, 0.0, 70000, "FSK", 20)
    # misaligned device pointer
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 20).to_c(np.float32)
    out = torch.empty(100, dtype=torch.float32, device="cuda")
    assert lib.urhgpu_afp_demod_dev(h, C.c_void_p(iq.data_ptr() + 8), 100, C.byref(p), C.c_void_p(out.data_ptr())) == _lib.ERR_ARG


def test_capacity_regrowth_on_noise_capture(pipe, oracle):
    """a noise-only capture yields ~n/10 pulse-table rows, far beyond the default capacity of 4 rows per symbol:
    iq_to_bits_checked notices and repeats the pass with enough room"""
    import torch
    from urh_amd.pipeline import DemodParams
    rng = np.random.default_rng(8)
    n = 400_000
    iq = (0.3 * rng.standard_normal((n, 2))).astype(np.float32)
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1, 1000, 0.1, 8, True)
    qad = oracle.afp_demod(iq, 0.0, "FSK", 2)
    pp = oracle.grab_pulse_lens(qad, 0.0, 1, "FSK", 1000, 1, 1.0)
    assert len(pp) > 4 * (n // 1000) + 4096
    res = pipe.iq_to_bits_checked(torch.from_numpy(iq).cuda(), p)
    assert np.array_equal(res.ppseq(), pp)
    fb = oracle.ppseq_to_bits_flat(pp, 1000, 1, True, 8)
    assert all(np.array_equal(a, b) for a, b in zip(fb, res.flat()))


# ---- row 6b: spectrogram band-pass (Filter.apply_bandpass_filter) -- floating point with the tolerance of
# tests/test_bandpass_host.py: 2^-40 * sum|h| * max|x| in the reference's np.convolve regime, 2^-18 in its (single precision)
# FFT regime -------------------------------------------------------------------------------------------------------------
def test_bandpass_equals_reference_outputs():
    from test_bandpass_host import golden_cases, tolerance
    from urh_amd import filter as uf
    for name, x, (lo, hi, bw), y in golden_cases():
        got = uf.apply_bandpass_filter(x, lo, hi, bw)
        assert got.dtype == np.complex128 and got.shape == y.shape, name
        assert np.max(np.abs(got - y)) <= tolerance(x, uf.bandpass_taps(lo, hi, bw)), name


def test_bandpass_kat():
    """/root/reference/tests/test_filter.py:124-132: swapped band edges give the same result"""
    from urh_amd import filter as uf
    sig = (np.sin(2 * np.pi * 0.2 * np.arange(100)) + np.sin(2 * np.pi * 0.3 * np.arange(100))).astype(np.complex64)
    assert np.array_equal(uf.apply_bandpass_filter(sig, 0.1, 0.2), uf.apply_bandpass_filter(sig, 0.2, 0.1))


def test_bandpass_device_equals_model(pipe):
    """device entry point on random captures: every tap-count residue of the register window, tile edges, complex128 and
    fused complex64 output, and a capture cut in three with the neighbours' edges as halos (what a shard sees)"""
    import torch
    from test_bandpass_host import model_convolve
    from urh_amd import filter as uf
    rng = np.random.default_rng(8)
    # (from 128 taps and 4096 outputs on: overlap-save through the LDS FFT, 4096 points up to 1025 taps, 8192 up to 4097)
    for n, m in [(1, 1), (5, 3), (1023, 7), (1024, 8), (1025, 9), (4097, 51), (4096, 127), (4096, 128), (20_000, 401), (3000, 4001), (70_000, 1001),
                 (30_000, 1025), (30_000, 1026), (60_000, 4001), (50_000, 4097), (20_000, 4098)]:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        h = rng.standard_normal(m) + 1j * rng.standard_normal(m)
        shift, n_out = (min(n, m) - 1) // 2, max(n, m)
        want = model_convolve(x, h, shift, n_out)
        tol = float(np.sum(np.abs(h)) * np.max(np.abs(x))) * 2.0 ** -40
        d_x = torch.from_numpy(x).to(pipe.device)
        got = uf.convolve_dev(pipe, d_x, h, shift, n_out, out_complex64=False).cpu().numpy()
        assert np.max(np.abs(got - want)) <= tol, (n, m)
        got32 = uf.convolve_dev(pipe, d_x, h, shift, n_out, out_complex64=True).cpu().numpy()
        assert got32.dtype == np.complex64 and np.array_equal(got32, got.astype(np.complex64)), (n, m)
    n, m = 50_000, 301
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    h = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    shift = (m - 1) // 2
    d_x = torch.from_numpy(x).to(pipe.device)
    whole = uf.convolve_dev(pipe, d_x, h, shift, n, out_complex64=False).cpu().numpy()
    cuts = [0, 17_001, 33_333, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        left = d_x[max(0, a - (m - 1 - shift)):a]
        right = d_x[b:min(n, b + shift)]
        part = uf.convolve_dev(pipe, d_x[a:b].contiguous(), h, shift, b - a, out_complex64=False, left=left, right=right).cpu().numpy()
        # (the shards' FFT blocks start elsewhere than the whole capture's: equal to rounding, not bit for bit)
        assert np.max(np.abs(part - whole[a:b])) <= float(np.sum(np.abs(h)) * np.max(np.abs(x))) * 2.0 ** -40, (a, b)
    m = 51                                                     # the direct form (short filters) is position independent: bit for bit
    h = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    shift = (m - 1) // 2
    whole = uf.convolve_dev(pipe, d_x, h, shift, n, out_complex64=False).cpu().numpy()
    for a, b in zip(cuts[:-1], cuts[1:]):
        left = d_x[max(0, a - (m - 1 - shift)):a]
        right = d_x[b:min(n, b + shift)]
        part = uf.convolve_dev(pipe, d_x[a:b].contiguous(), h, shift, b - a, out_complex64=False, left=left, right=right).cpu().numpy()
        assert np.array_equal(part, whole[a:b]), (a, b)


def test_bandpass_argument_errors(pipe):
    import torch
    from urh_amd import _lib, filter as uf
    x = torch.zeros(100, dtype=torch.complex64, device=pipe.device)
    with pytest.raises(_lib.UrhGpuError):
        uf.convolve_dev(pipe, x, np.ones(40_000, dtype=np.complex128), 0, 100)        # more taps than the LDS window holds
    with pytest.raises(ValueError):
        uf.apply_bandpass_filter(np.zeros(0, np.complex64), 0.1, 0.2)


def test_shard_result_host_looked_at_too_late_raises():
    """Three blob slots rotate (GpuShardEngine(host_results=True)): a result whose host() is asked for after three later passes have been
    issued would read another pass's blob -- it raises instead; results looked at in time are unaffected."""
    import torch
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import ShardedPipeline, ThreadComm
    n = 300_000
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 50, 0.1, 8, True)
    caps = [torch.from_numpy(synth_fsk(n, sps=50, seed=70 + i, noise=0.05)).cuda() for i in range(5)]
    sp = ShardedPipeline(GpuShardEngine(0, pipelined=True, host_results=True), ThreadComm(ThreadComm.Shared(1), 0))
    res = [sp.iq_to_bits(c, p, want_qad=True) for c in caps]
    with pytest.raises(RuntimeError, match="reused"):
        res[0].host()
    with pytest.raises(RuntimeError, match="reused"):
        res[1].host()
    for k in (2, 3, 4):
        h = res[k].host().check()
        assert np.array_equal(h.ppseq(), res[k].piece()["rows"]) if k == 4 else h.n_rows > 0
