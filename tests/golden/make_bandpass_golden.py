#!/usr/bin/env python3
"""Run the REAL reference's Filter.apply_bandpass_filter (this container only) on the first samples of golden captures and
on synthetic tones, in both of its regimes (np.convolve for short filters, FFT convolution for long ones), and store
inputs, parameters and the complex128 results in tests/golden/filter/bandpass.npz; the GPU test compares
urh_amd.filter.apply_bandpass_filter with it within the tolerance stated there.

    python tests/golden/make_bandpass_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_python  # noqa: E402

ref_python.setup()
from urh.signalprocessing.Filter import Filter  # noqa: E402

rng = np.random.default_rng(20260923)
cases = []
fsk = np.load(os.path.join(HERE, "fsk.npz"))["iq"]
fsk = (fsk[:6000, 0] + 1j * fsk[:6000, 1]).astype(np.complex64)
tone = (np.exp(2j * np.pi * 0.12 * np.arange(3000)) + 0.5 * np.exp(-2j * np.pi * 0.31 * np.arange(3000))).astype(np.complex64)
noise = (rng.standard_normal(40) + 1j * rng.standard_normal(40)).astype(np.complex64)
for name, x, lo, hi, bw in [
    ("fsk_default", fsk, -0.05, 0.08, 0.08),          # 51 taps >= 8 ln sqrt(6000) = 34.8: FFT path
    ("fsk_wide", fsk, 0.3, -0.4, 0.2),                # 21 taps, swapped edges: np.convolve path
    ("fsk_narrow", fsk, 0.01, 0.02, 0.004),           # 1001 taps: FFT path
    ("tone_pos", tone, 0.1, 0.14, 0.05),              # 81 taps
    ("tone_neg_clip", tone, -0.9, -0.25, 0.3),        # 15 taps, f_low clipped to -0.5: np.convolve path
    ("short_capture", noise, 0.0, 0.2, 0.08),         # 51 taps on 40 samples: more taps than samples
]:
    y = np.asarray(Filter.apply_bandpass_filter(x, lo, hi, filter_bw=bw))
    assert y.dtype == np.complex128
    cases.append((name, x, np.array([lo, hi, bw]), y))
    print(name, len(x), Filter.get_filter_length_from_bandwidth(bw), len(y))
np.savez_compressed(os.path.join(HERE, "filter", "bandpass.npz"), names=np.array([c[0] for c in cases]),
                    **{f"x_{c[0]}": c[1] for c in cases}, **{f"p_{c[0]}": c[2] for c in cases}, **{f"y_{c[0]}": c[3] for c in cases})
