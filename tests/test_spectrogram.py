"""Spectrogram on the GPU (spectrogram.hip) against arrays produced by the REAL reference class
(tests/golden/spectrogram/spectrogram.npz, made by tests/golden/make_spectrogram_golden.py).  Floating point: the FFT runs
in double precision like numpy's but with radix-2 butterflies instead of pocketfft's, so
    stft        |difference| <= 1e-12 (values are O(1) after the division by window_size)
    decibels    |difference| <= 1e-4 dB where the reference is above -200 dB (a complex64 rounding flip moves a bin by
                5e-7 dB, log10f implementations differ by ~1e-5 dB); bins below that are rounding noise of the FFT itself in both
                implementations and only have to be "nothing there" (<= -190 dB); -inf (exact zeros) must be -inf
    image       identical BGRA bytes except for bins that sit on a colormap step (<= 0.1 % of the pixels)
"""
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "spectrogram", "spectrogram.npz")


def cases():
    g = np.load(GOLD)
    for name in g["names"]:
        name = str(name)
        ws, ov = g[name + "_args"]
        yield name, g[name + "_x"], int(ws), float(ov), g[name + "_stft"], g[name + "_db"], g[name + "_img"], g["colormap"]


def test_golden_file_is_what_the_mirror_expects():
    n = 0
    for name, x, ws, ov, st, db, img, cm in cases():
        hop = ws - int(ov * ws)
        frames = max(1, (max(len(x), ws) - ws) // hop + 1)
        assert st.shape == (frames, ws) and st.dtype == np.complex128 and db.shape == (frames, ws) and db.dtype == np.float32
        assert img.shape == (ws, frames, 4) and img.dtype == np.uint8 and cm.shape[1] == 4
        n += 1
    assert n == 5


def check_db(got, want, name):
    assert got.shape == want.shape and got.dtype == np.float32
    ninf = np.isneginf(want)
    assert np.array_equal(np.isneginf(got), ninf), name
    loud = want > -200
    assert np.max(np.abs(got[loud] - want[loud]), initial=0.0) <= 1e-4, name
    quiet = ~loud & ~ninf
    assert (got[quiet] <= -190).all(), name


@pytest.mark.gpu
def test_gpu_spectrogram_equals_reference():
    import torch
    from urh_amd.spectrogram import Spectrogram
    for name, x, ws, ov, st, db, img, cm in cases():
        sp = Spectrogram(x, window_size=ws, overlap_factor=ov)
        assert np.max(np.abs(sp.stft() - st)) <= 1e-12, name
        check_db(sp.calculate_spectrogram(), db, name)
        dev = Spectrogram(torch.from_numpy(x).cuda(), window_size=ws, overlap_factor=ov)       # capture already in HBM
        check_db(dev.calculate_spectrogram(device=True).cpu().numpy(), db, name + " (device)")
        with np.errstate(all="ignore"):
            got = sp.apply_bgra_lookup(db, cm, sp.data_min, sp.data_max)                       # the lookup alone: exact
        assert np.array_equal(got, img), name
        full = sp.create_spectrogram_image_array(cm)
        assert full.shape == img.shape and np.mean(np.any(full != img, axis=2)) <= 1e-3, name


@pytest.mark.gpu
def test_gpu_spectrogram_large_and_window_sizes():
    """2^20 samples, every supported window size: against numpy's FFT of the same frames (the reference's formula)."""
    from urh_amd.spectrogram import Spectrogram
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20)).astype(np.complex64)
    for ws in (8, 64, 1024, 4096):
        sp = Spectrogram(x[: 1 << (12 + (ws > 64) * 8)], window_size=ws)
        xs = sp.samples
        hop = sp.hop_size
        frames = (len(xs) - ws) // hop + 1
        idx = np.arange(ws)[None, :] + hop * np.arange(frames)[:, None]
        ref = np.fft.fft(xs[idx] * np.hanning(ws), ws) / ws
        assert np.max(np.abs(sp.stft() - ref)) <= 1e-12, ws
        want = np.fliplr((10 * np.log10(np.abs(np.fft.fftshift(ref, axes=(1,)).astype(np.complex64)) ** 2)).astype(np.float32))
        got = sp.calculate_spectrogram()
        ok = want > -200
        assert np.max(np.abs(got[ok] - want[ok])) <= 1e-3, ws
    with pytest.raises(Exception):
        Spectrogram(x[:5000], window_size=1000).stft()          # not a power of two: URHGPU_ERR_UNSUPPORTED
