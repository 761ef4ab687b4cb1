"""urh_amd.auto_interpretation / urh_amd.util -- the drop-in mirrors of urh.cythonext.auto_interpretation / util -- against
the REAL reference functions (oracle/_ref = the reference's Cython modules compiled here) on randomised inputs.  The two functions
that are native host arithmetic in the library (get_threshold_divisor_histogram, merge_plateaus) are checked without a GPU."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref():
    import build_ref
    if not build_ref.built():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    sf, util, ai = build_ref.import_ref()
    return sf, util, ai


def test_divisor_histogram_and_merge_equal_reference(ref):
    _, _, c_ai = ref
    from urh_amd import auto_interpretation as mine
    rng = np.random.default_rng(11)
    for it in range(300):
        n = int(rng.integers(1, 300))
        base = int(rng.choice([1, 7, 40, 100]))
        p = (base * rng.integers(0, 7, n) + rng.integers(0, 4, n)).astype(np.uint64)
        for thr in (0.2, 0.05, 0.5, 0.0):
            want = np.asarray(c_ai.get_threshold_divisor_histogram(p.copy(), thr))
            got = mine.get_threshold_divisor_histogram(p.copy(), thr)
            assert got.dtype == np.uint64 and np.array_equal(want, got), (it, thr)
        for tol, mc in ((0, 10000), (2, 10000), (5, 3), (1, 1), (50, 10000)):
            want = np.asarray(c_ai.merge_plateaus(p.copy(), tol, mc))
            got = mine.merge_plateaus(p.copy(), tol, mc)
            assert got.dtype == np.uint64 and np.array_equal(want, got), (it, tol, mc)
    assert len(mine.merge_plateaus(np.zeros(0, np.uint64), 1, 10)) == 0
    with pytest.raises(ValueError):
        mine.get_threshold_divisor_histogram(np.zeros(0, np.uint64))


def _bursty_magnitudes(rng, n):
    m = np.abs(rng.standard_normal(n)) * 0.02
    pos = 0
    while pos < n:
        gap = int(rng.integers(1, 400))
        burst = int(rng.integers(1, 600))
        m[pos + gap:pos + gap + burst] += 0.5
        pos += gap + burst
    drop = rng.random(n) < 0.03                                      # outliers inside bursts and gaps
    m[drop] = rng.random(int(drop.sum())) * 0.6
    return m


@pytest.mark.gpu
def test_segment_messages_from_magnitudes_equals_reference(ref):
    _, _, c_ai = ref
    from urh_amd import auto_interpretation as mine
    rng = np.random.default_rng(3)
    for it in range(40):
        n = int(rng.choice([1, 5, 9, 10, 11, 57, 3000, 40_000, 300_000]))
        m64 = _bursty_magnitudes(rng, n)
        if it % 5 == 1:
            m64[:12] = 0.7                                           # starts above the noise
        if it % 5 == 2:
            m64[-15:] = 0.7                                          # ends above the noise
        if it % 5 == 3:
            m64[-4:] = 0.0
            m64[-30:-4] = 0.7                                        # trailing below-run shorter than the tolerance
        for dt in (np.float64, np.float32):
            m = m64.astype(dt)
            thr = 0.25
            assert mine.segment_messages_from_magnitudes(m, thr) == c_ai.segment_messages_from_magnitudes(m, thr), (it, n, dt)
        # a threshold that is not representable in float32 compared with float64 magnitudes (the reference promotes the C float)
        m = m64.copy()
        m[::7] = float(np.float32(0.1))
        assert mine.segment_messages_from_magnitudes(m, 0.1) == c_ai.segment_messages_from_magnitudes(m, 0.1), it
    assert mine.segment_messages_from_magnitudes(np.zeros(0, np.float32), 0.1) == []


@pytest.mark.gpu
def test_get_plateau_lengths_and_median_filter_equal_reference(ref):
    _, _, c_ai = ref
    from urh_amd import auto_interpretation as mine
    rng = np.random.default_rng(5)
    for it in range(40):
        n = int(rng.choice([1, 3, 4, 100, 5000, 200_000]))
        sps = int(rng.choice([1, 7, 100, 3000]))
        x = (np.repeat(rng.integers(0, 2, n // sps + 1), sps)[:n] * 0.8 + 0.02 * rng.standard_normal(n)).astype(np.float32)
        if it % 7 == 0:
            x[:] = 1.0                                               # no boundary at all
        for pct in (25, 0, 100, 60):
            want = np.asarray(c_ai.get_plateau_lengths(x, 0.4, pct))
            got = mine.get_plateau_lengths(x, 0.4, pct)
            assert got.dtype == np.uint64 and np.array_equal(want, got), (it, n, sps, pct)
        d = rng.standard_normal(min(n, 20_000)) * 1e3
        for k in (3, 11, 1, 4, 64):
            want = np.asarray(c_ai.median_filter(d, k))
            got = mine.median_filter(d, k=k)
            assert got.dtype == np.float32 and np.array_equal(want.view(np.uint32), got.view(np.uint32)), (it, k)
    assert len(mine.get_plateau_lengths(np.zeros(0, np.float32), 0.5)) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.float32])
def test_minmax_equals_reference(ref, dtype):
    _, c_util, _ = ref
    from urh_amd import util as mine
    rng = np.random.default_rng(9)
    for n in (1, 2, 63, 64, 65, 1000, 300_001):
        if np.dtype(dtype) == np.float32:
            a = rng.standard_normal(n).astype(np.float32)
        else:
            info = np.iinfo(dtype)
            a = rng.integers(info.min, info.max + 1, n).astype(dtype)
        assert mine.minmax(a) == c_util.minmax(a), (dtype, n)
    assert mine.minmax(np.zeros(0, dtype)) == (0, 0)
    if np.dtype(dtype) == np.float32:                               # NaN: skipped by the comparisons unless it is element 0
        a = rng.standard_normal(5000).astype(np.float32)
        a[100] = np.nan
        a[4000] = np.inf
        assert mine.minmax(a) == c_util.minmax(a)
        a[0] = np.nan
        got, want = mine.minmax(a), c_util.minmax(a)
        assert np.isnan(got[0]) and np.isnan(got[1]) and np.isnan(want[0]) and np.isnan(want[1])
