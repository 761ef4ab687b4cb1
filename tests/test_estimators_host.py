"""Host-side decision logic of urh_amd/estimators.py (no GPU involved) against the reference's own functions
(/root/reference: AutoInterpretation.py, auto_interpretation.pyx) on randomised plateau lengths / segments.
Needs the oracle/_ref build; skipped where /root/reference is absent."""
import numpy as np
import pytest

from urh_amd import estimators as e


@pytest.fixture(scope="module")
def ref():
    import build_ref
    import ref_python
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("reference Python not available")
    ref_python.setup()
    from urh.ainterpretation import AutoInterpretation as AI
    from urh.cythonext import auto_interpretation as c_ai
    return AI, c_ai


def _plateaus(rng):
    n = int(rng.integers(0, 400))
    base = int(rng.choice([8, 40, 100, 295]))
    p = base * rng.integers(1, 6, n) + rng.integers(-3, 4, n)
    glitch = rng.random(n) < rng.choice([0.0, 0.1, 0.3])
    p[glitch] = rng.integers(1, 4, int(glitch.sum()))
    return np.maximum(p, 1).astype(np.uint64)


def test_plateau_logic_equals_reference(ref):
    AI, c_ai = ref
    rng = np.random.default_rng(0)
    for it in range(400):
        p = _plateaus(rng)
        tol_ref = AI.estimate_tolerance_from_plateau_lengths(p)
        tol = e.estimate_tolerance_from_plateau_lengths(p)
        assert (tol_ref is None and tol is None) or int(tol_ref) == int(tol), (it, tol_ref, tol)
        for t in (0, 1, 3, 5, None):
            a = AI.merge_plateau_lengths(p.copy(), tolerance=t)
            b = e.merge_plateau_lengths(p.copy(), tolerance=t)
            assert np.array_equal(np.asarray(a), np.asarray(b)), (it, t)
        if len(p):
            assert np.array_equal(np.asarray(c_ai.get_threshold_divisor_histogram(p.copy())), e.get_threshold_divisor_histogram(p.copy())), it
            assert np.array_equal(np.asarray(c_ai.merge_plateaus(p.copy(), 2, 50)), e.merge_plateaus(p.copy(), 2, 50)), it
        a, b = p.copy(), p.copy()
        assert AI.get_bit_length_from_plateau_lengths(a) == e.get_bit_length_from_plateau_lengths(b), it
        assert np.array_equal(a, b)                              # rounded in place the same way


def test_segment_and_value_helpers_equal_reference(ref):
    AI, _ = ref
    rng = np.random.default_rng(1)
    for it in range(300):
        n = int(rng.integers(0, 30))
        starts = np.cumsum(rng.integers(5, 4000, n))
        segs = [(int(s), int(s + rng.integers(1, 3000))) for s in starts]
        segs = [s for i, s in enumerate(segs) if i == 0 or s[0] > segs[i - 1][1]] if segs else segs
        segs2 = []
        for s in segs:
            if not segs2 or s[0] > segs2[-1][1]:
                segs2.append(s)
        a = AI.merge_message_segments_for_ook(list(segs2))
        b = e.merge_message_segments_for_ook(list(segs2))
        assert [tuple(map(int, x)) for x in a] == [tuple(map(int, x)) for x in b], (it, segs2)
        from oracle import urh_oracle
        c = urh_oracle.merge_message_segments_for_ook(list(segs2))
        assert [tuple(map(int, x)) for x in a] == [tuple(map(int, x)) for x in c], (it, segs2)
        vals = [int(v) for v in rng.integers(0, 6, int(rng.integers(0, 12)))]
        assert AI.get_most_frequent_value(list(vals)) == e.get_most_frequent_value(list(vals)), vals
        d = rng.standard_normal(int(rng.integers(0, 20)))
        for fn in ("max_without_outliers", "min_without_outliers"):
            x, y = getattr(AI, fn)(d), getattr(e, fn)(d)
            assert (x is None and y is None) or x == y


def test_median_filter_equals_reference(ref):
    AI, c_ai = ref
    rng = np.random.default_rng(4)
    for n in (0, 1, 2, 5, 10, 11, 12, 100, 1000):
        x = rng.standard_normal(n) * rng.choice([1e-3, 1.0, 1e6])
        for k in (1, 2, 3, 11, 12):
            want = c_ai.median_filter(x, k=k)
            got = e.median_filter(x, k=k)
            assert got.dtype == want.dtype and np.array_equal(got, want), (n, k)


def test_detect_modulation_equals_reference(ref):
    """AutoInterpretation.detect_modulation on every message the reference segments out of the golden captures, plus
    synthetic ASK / PSK / noise messages: the same label."""
    import os
    from conftest import GOLDEN_DIR
    AI, c_ai = ref
    from urh.signalprocessing.IQArray import IQArray
    n_checked = 0
    for f in sorted(os.listdir(GOLDEN_DIR)):
        if not f.endswith(".npz"):
            continue
        z = np.load(os.path.join(GOLDEN_DIR, f))
        if "iq" not in z:
            continue
        iq = z["iq"]
        arr = IQArray(iq)
        noise = AI.detect_noise_level(arr.magnitudes)
        segs = AI.segment_messages_from_magnitudes(arr.magnitudes, noise_threshold=noise)
        cplx = arr.as_complex64()
        assert np.array_equal(e._as_complex64(iq).view(np.uint32), cplx.view(np.uint32)), f
        for start, end in segs[:100]:
            assert e.detect_modulation(cplx[start:end]) == AI.detect_modulation(cplx[start:end]), (f, start, end)
            n_checked += 1
    rng = np.random.default_rng(12)
    t = np.arange(4096)
    for msg in (np.exp(2j * np.pi * 0.05 * t) * np.repeat(rng.integers(1, 3, 64), 64),
                np.exp(2j * np.pi * 0.05 * t + 1j * np.pi * np.repeat(rng.integers(0, 2, 64), 64)),
                rng.standard_normal(4096) + 1j * rng.standard_normal(4096), np.zeros(100), np.ones(3)):
        msg = (msg + 0.01 * (rng.standard_normal(len(msg)) + 1j * rng.standard_normal(len(msg))) * (np.abs(msg) > 0)).astype(np.complex64)
        assert e.detect_modulation(msg) == AI.detect_modulation(msg)
        n_checked += 1
    assert n_checked > 20
