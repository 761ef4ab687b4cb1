"""Executable model of fft_peak.hip's index arithmetic (CPU): the Stockham autosort stage the row kernel runs in LDS and the four-step
factorisation n = n1 n2 (transpose, row FFTs + twiddle, transpose, row FFTs; X[k1 + n1 k2] = E[k1][k2]) against numpy's FFT -- the same
loops as the kernel's, in numpy, so that a change of the kernel's indexing has a checker that runs without a GPU
(Signal.estimate_frequency, Signal.py:578-601)."""
import numpy as np


def stockham(x):
    L = len(x)
    src, dst = x.astype(np.complex64).copy(), np.empty(L, np.complex64)
    tw = np.exp(-2j * np.pi * np.arange(max(L // 2, 1)) / L).astype(np.complex64)
    n, s = L, 1
    while n > 1:
        m = n // 2
        idx = np.arange(L // 2)
        p, q = idx // s, idx % s
        a, b = src[q + s * p], src[q + s * (p + m)]
        dst[q + s * (2 * p)] = a + b
        dst[q + s * (2 * p + 1)] = (a - b) * tw[p * (L // n)]
        src, dst = dst, src
        n, s = m, s * 2
    return src


def four_step(x):
    n = len(x)
    k = int(np.log2(n))
    n1 = 1 << ((k + 1) // 2)
    n2 = n // n1
    b = x.reshape(n1, n2).T.copy()                                         # B[j2][j1]
    c = np.stack([stockham(row) for row in b])                               # C[j2][k1]
    j2, k1 = np.arange(n2)[:, None], np.arange(n1)[None, :]
    c = c * np.exp(-2j * np.pi * ((j2 * k1) % n) / n).astype(np.complex64)
    e = np.stack([stockham(row) for row in c.T.copy()])                      # E[k1][k2]
    i = np.arange(n)
    out = np.empty(n, np.complex64)
    out[(i // n2) + (i % n2) * n1] = e.reshape(-1)                           # the argmax kernel's k = (e >> log2 n2) + ((e & (n2 - 1)) << log2 n1)
    return out


def test_stockham_stage_indexing_and_four_step_mapping():
    rng = np.random.default_rng(0)
    for L in (1, 2, 4, 8, 64, 1024, 8192):
        x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.max(np.abs(stockham(x) - ref)) <= 1e-6 * max(1.0, np.max(np.abs(ref))), L
    for n in (16, 32, 2048, 1 << 15):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.max(np.abs(four_step(x) - ref)) <= 1e-6 * np.max(np.abs(ref)), n
