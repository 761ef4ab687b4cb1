"""Pin the oracle: against the committed golden vectors (produced by the real reference, see
tests/golden/make_golden.py), against the reference's own known-answer bit strings, and -- where the
oracle/_ref build of the real Cython modules is present -- against the reference itself on random input."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, load_golden, synth_fsk


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_matches_golden(oracle, name):
    g = load_golden(name)
    mod, bps = g["modulation_type"], g["bits_per_symbol"]
    qad = oracle.afp_demod(g["iq"], g["noise_threshold"], mod, 2 ** bps, g["costas_loop_bandwidth"])
    start = 1 if mod == "PSK" else 0          # reference leaves result[0] uninitialised for PSK
    assert np.array_equal(qad[start:].view(np.uint32), g["qad"][start:].view(np.uint32))
    pp = oracle.grab_pulse_lens(g["qad"], g["center"], g["tolerance"], mod, g["samples_per_symbol"], bps, g["center_spacing"])
    assert np.array_equal(pp, g["ppseq"])
    bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(g["ppseq"], g["samples_per_symbol"], bps, True, g["pause_threshold"])
    assert np.array_equal(bits, g["bits"]) and np.array_equal(off, g["msg_off"])
    assert np.array_equal(pauses, g["pauses"]) and np.array_equal(pos, g["pos"]) and np.array_equal(poff, g["pos_off"])
    if g["kat"]:
        first = "".join(map(str, bits[off[0]:off[1]]))
        assert first == g["kat"] if g["kat_mode"] == "exact" else first.startswith(g["kat"])


def test_fir_kat(oracle):
    """/root/reference/tests/test_filter.py:20-31"""
    x = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 42], dtype=np.complex64)
    out = oracle.fir_filter(x, np.array([0.25] * 4, dtype=np.complex64))
    assert np.allclose(out, [0.25, 0.75, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 16.5])


def _ref():
    import build_ref
    if not build_ref.built():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return build_ref.import_ref()


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.uint8, np.int16, np.uint16])
def test_oracle_vs_real_reference_random(oracle, dtype):
    sf, util, ai = _ref()
    rng = np.random.default_rng(7)
    iq = synth_fsk(30000, sps=50, seed=3, noise=0.08, pause_every=7000, pause_len=900, dtype=dtype)
    assert np.array_equal(util.get_magnitudes(iq), oracle.get_magnitudes(iq), equal_nan=True)   # uint16: C int overflow -> NaN, as in the reference
    scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64)).max())
    for mod in ("ASK", "FSK", "PSK"):
        for noise in (0.0, 0.3 * scale):
            for order in (2, 4):
                a = np.asarray(sf.afp_demod(iq, noise, mod, order))
                b = oracle.afp_demod(iq, noise, mod, order)
                st = 1 if mod == "PSK" else 0
                assert np.array_equal(a[st:].view(np.uint32), b[st:].view(np.uint32)), (mod, noise, order)
                a[0] = b[0]
                for tol in (0, 1, 5, 60):
                    for bps in (1, 2):
                        c = float(rng.uniform(-0.1, 0.4))
                        r1 = np.asarray(sf.grab_pulse_lens(a, c, tol, mod, 50, bps, 0.2))
                        r2 = oracle.grab_pulse_lens(b, c, tol, mod, 50, bps, 0.2)
                        assert np.array_equal(r1, r2), (mod, noise, tol, bps)


def test_oracle_vs_real_reference_edge_values(oracle):
    """signed zeros, infinities, NaNs and denormals through the FSK conj-product / atan2f path"""
    sf, util, ai = _ref()
    vals = np.array([0.0, -0.0, 1.0, -1.0, 1e-40, -1e-40, 1e-30, 3e38, -3e38, 0.5, -0.25, np.inf, -np.inf, np.nan],
                    dtype=np.float32)
    rng = np.random.default_rng(5)
    iq = vals[rng.integers(0, len(vals), size=(40000, 2))]
    for mod in ("FSK", "ASK"):
        a = np.asarray(sf.afp_demod(iq, 0.0, mod, 2))
        b = oracle.afp_demod(iq, 0.0, mod, 2)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), (mod, int((~same).sum()))


def test_oracle_filters_vs_real_reference(oracle):
    sf, util, ai = _ref()
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64)
    h = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype(np.complex64)
    assert np.array_equal(np.asarray(sf.fir_filter(x, h)).view(np.uint32), oracle.fir_filter(x, h).view(np.uint32))
    a = rng.standard_normal(3) * 0.3
    b = rng.standard_normal(2) * 0.3
    assert np.array_equal(np.asarray(sf.iir_filter(a, b, x)).view(np.uint32), oracle.iir_filter(a, b, x).view(np.uint32))


def _filter_goldens():
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "filter", "fir_iir.npz"))
    return g, [str(n) for n in g["names"]]


def test_oracle_filters_equal_reference_goldens(oracle):
    """fir_filter / iir_filter outputs of the real reference (tests/golden/make_filter_golden.py): the committed vectors are
    the pin for iir_filter, which the reference's own tests never assert on."""
    g, names = _filter_goldens()
    assert len(names) == 9
    for name in names:
        if name.startswith("iir"):
            got = oracle.iir_filter(g[name + "_a"], g[name + "_b"], g[name + "_x"])
        else:
            got = oracle.fir_filter(g[name + "_x"], g[name + "_h"])
        assert np.array_equal(got.view(np.uint32), g[name + "_y"].view(np.uint32)), name


def test_oracle_ppseq_to_bits_vs_reference_python(oracle):
    """_ppseq_to_bits against the reference's own Python (needs /root/reference + the PyQt6 stub)."""
    import ref_python
    import build_ref
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("reference Python not available")
    ref_python.setup()
    from urh.signalprocessing.ProtocolAnalyzer import ProtocolAnalyzer
    pa = ProtocolAnalyzer(None)
    rng = np.random.default_rng(11)
    for it in range(200):
        n = int(rng.integers(0, 60))
        lens = rng.integers(-5, 900, n)
        lens[rng.random(n) < 0.2] *= 13
        sps, bps = int(rng.choice([1, 7, 50, 100])), int(rng.choice([1, 2]))
        types = rng.integers(-1, 2 ** bps, n)          # valid states only: -1 .. 2^bps - 1
        pp = np.stack([types, lens], axis=1).astype(np.int64)
        for pt in (0, 1, 8):
            ref = pa._ppseq_to_bits(pp, sps, bps, pause_threshold=pt)
            got = oracle.ppseq_to_bits(pp, sps, bps, True, pt)
            assert [list(x) for x in ref[0]] == [list(x) for x in got[0]]
            assert list(ref[1]) == list(got[1])
            assert [list(x) for x in ref[2]] == [list(x) for x in got[2]]


def test_oracle_estimators_vs_reference_python(oracle):
    """The numpy restatements of the estimators (detect_noise_level, detect_center, segment_messages_from_magnitudes,
    get_plateau_lengths) against the reference's own AutoInterpretation / Cython code, on every golden capture and on
    seeded synthetic input."""
    import ref_python
    import build_ref
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("reference Python not available")
    ref_python.setup()
    from urh.ainterpretation import AutoInterpretation as AI
    from urh.cythonext import auto_interpretation as c_ai
    cases = [(name, load_golden(name)) for name in GOLDEN_CASES]
    for name, g in cases:
        mags = oracle.get_magnitudes(g["iq"])
        assert AI.detect_noise_level(mags) == oracle.detect_noise_level(mags), name
        a, b = AI.detect_center(g["qad"]), oracle.detect_center(g["qad"])
        assert (a is None and b is None) or float(a) == float(b), (name, a, b)
        nt = float(g["noise_threshold"]) or 0.01
        m = mags[:60000]
        assert [tuple(map(int, s)) for s in AI.segment_messages_from_magnitudes(m, nt)] == \
            [tuple(map(int, s)) for s in oracle.segment_messages_from_magnitudes(m, nt)], name
        if a is not None:
            q = np.ascontiguousarray(g["qad"][:40000])
            assert np.array_equal(np.asarray(c_ai.get_plateau_lengths(q, a, 25)), oracle.get_plateau_lengths(q, a, 25)), name
    for n, seed in ((5000, 1), (100_003, 2)):
        iq = synth_fsk(n, sps=50, seed=seed, noise=0.05, pause_every=n // 3, pause_len=n // 11)
        qad = oracle.afp_demod(iq, 0.2, "FSK", 2)
        a, b = AI.detect_center(qad), oracle.detect_center(qad)
        assert (a is None and b is None) or float(a) == float(b), (n, a, b)
        a, b = AI.detect_center(qad, max_size=3000), oracle.detect_center(qad, max_size=3000)
        assert (a is None and b is None) or float(a) == float(b), (n, "max_size", a, b)


def test_config1_reference_plumbing_cpu():
    """BASELINE.json configs[0]: the reference's own headless hot-path tests (tests/test_demodulations.py with the
    tests/data FSK capture first of all) on the compiled Cython path, CPU only -- the same driver tests/test_reference_dropin.py runs
    on the GPU box with liburhgpu.so bound underneath."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_driver.py"), "--no-patch"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if out.get("unavailable"):
        pytest.skip("oracle/_ref not built")
    assert out["per_module"]["tests.test_demodulations"]["ran"] == 7
    assert out["ran"] == 75 and out["failures"] == 0 and out["errors"] == 0, out["details"]
    assert out["calls"]["signal_functions.afp_demod"] >= 8 and out["calls"]["signal_functions.grab_pulse_lens"] >= 8


def test_hook_keeps_cython_without_gpu():
    """The reference-side hook (urh_amd/urh_hook.py) on a host WITHOUT a GPU: it probes
    urhgpu_ctx_create, says why the library cannot be used and leaves URH's Cython functions bound -- the reference's 75 headless hot-path
    tests then run green on the Cython path with the hook installed.  (On a GPU box the same flag binds the library:
    tests/test_reference_dropin.py::test_hook_binds_on_a_gpu_box.)"""
    import json
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the hook binds the library here (covered by the gpu test)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_driver.py"), "--hook"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if out.get("unavailable"):
        pytest.skip("oracle/_ref not built")
    assert out["hook"]["installed"] is False and out["hook"]["log"] and "keeping the Cython functions" in out["hook"]["log"][0], out["hook"]
    assert out["ran"] == 75 and out["failures"] == 0 and out["errors"] == 0, out["details"]
    assert all(v == 0 for v in out["calls"].values()), out["calls"]          # nothing went through the (unusable) library
