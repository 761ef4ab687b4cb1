"""Host side of the spectrogram band-pass (urh_amd/filter.py, row 6b): tap design and result geometry against the
reference's Filter (skipped where /root/reference is absent), and the centred-convolution formula the kernel evaluates
against the committed outputs of the real reference (tests/golden/filter/bandpass.npz).  No GPU involved."""
import os

import numpy as np
import pytest

from urh_amd import filter as f

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "filter", "bandpass.npz")


def model_convolve(x, h, shift, n_out, left=None, right=None):
    """out[i] = sum_k h[k] * X(i + shift - k) in complex128 with numpy (the formula of csrc/bandpass.hip)"""
    x = np.asarray(x, dtype=np.complex128)
    nl = 0 if left is None else len(left)
    ext = np.concatenate([np.zeros(0) if left is None else np.asarray(left, np.complex128), x,
                          np.zeros(0) if right is None else np.asarray(right, np.complex128)])
    full = np.convolve(ext, np.asarray(h, np.complex128), "full") if len(ext) and len(h) else np.zeros(0, np.complex128)
    out = np.zeros(n_out, dtype=np.complex128)
    # full[j + nl] = sum_k h[k] ext[j + nl - k] = Y(j); out[i] = Y(i + shift)
    j = np.arange(n_out) + shift + nl
    ok = (j >= 0) & (j < len(full))
    out[ok] = full[j[ok]]
    return out


def golden_cases():
    z = np.load(GOLDEN)
    return [(str(n), z[f"x_{n}"], z[f"p_{n}"], z[f"y_{n}"]) for n in z["names"]]


def tolerance(x, h):
    """|error| allowed per output, relative to the largest possible magnitude S = sum_k |h[k]| * max|x|:
    2^-40 * S where the reference uses np.convolve (complex128 dot products; only the summation order differs), and
    2^-18 * S where it uses the FFT convolution (Filter.py:70-82): numpy >= 2 transforms the complex64 capture in SINGLE
    precision there, so the reference itself is only float32-accurate in that regime (the caller casts the result to
    complex64 anyway, SignalFrame.py:1578-1580).  The kernel is fp64 throughout, i.e. closer to the exact result."""
    import math
    s = float(np.sum(np.abs(h)) * np.max(np.abs(x)))
    return s * (2.0 ** -40 if len(h) < 8 * math.log(math.sqrt(len(x))) else 2.0 ** -18)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c[0])
def test_formula_equals_reference_outputs(case):
    name, x, (lo, hi, bw), y = case
    h = f.bandpass_taps(lo, hi, bw)
    shift, n_out = f._same_geometry(len(x), len(h))
    assert n_out == len(y)
    got = model_convolve(x, h, shift, n_out)
    assert np.max(np.abs(got - y)) <= tolerance(x, h)


def test_filter_length():
    assert [f.get_filter_length_from_bandwidth(b) for b in (0.08, 0.1, 0.05, 0.3, 4.0, 0.004)] == [51, 41, 81, 15, 1, 1001]


@pytest.fixture(scope="module")
def ref_filter():
    import build_ref
    import ref_python
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("reference Python not available")
    ref_python.setup()
    from urh.signalprocessing.Filter import Filter
    return Filter


def test_taps_and_geometry_equal_reference(ref_filter):
    rng = np.random.default_rng(5)
    for _ in range(40):
        lo, hi = rng.uniform(-0.7, 0.7, 2)
        bw = float(rng.choice([0.3, 0.08, 0.05, 0.011]))
        n = int(rng.integers(1, 400))
        mine = f.bandpass_taps(lo, hi, bw)
        a, b = (hi, lo) if lo > hi else (lo, hi)
        theirs = ref_filter.design_windowed_sinc_bandpass(max(-0.5, min(a, 0.5)), max(-0.5, min(b, 0.5)), bw)
        assert mine.dtype == theirs.dtype and np.array_equal(mine, theirs)
        assert ref_filter.get_filter_length_from_bandwidth(bw) == f.get_filter_length_from_bandwidth(bw) == len(mine)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        y = ref_filter.apply_bandpass_filter(x, lo, hi, filter_bw=bw)
        shift, n_out = f._same_geometry(n, len(mine))
        assert n_out == len(y)
        assert np.max(np.abs(model_convolve(x, mine, shift, n_out) - y), initial=0.0) <= tolerance(x, mine)


def test_empty_capture_raises_like_the_reference():
    with pytest.raises(ValueError):
        f._same_geometry(0, 51)
