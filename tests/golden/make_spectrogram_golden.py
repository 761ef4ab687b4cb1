#!/usr/bin/env python3
"""Run the REAL reference's Spectrogram (oracle/_ref + PyQt6 stub, this container only): stft, the decibel array of
__calculate_spectrogram and apply_bgra_lookup through the reference's default colormap, on the start of golden captures
and on synthetic tones; results in tests/golden/spectrogram/spectrogram.npz.

    python tests/golden/make_spectrogram_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_python  # noqa: E402

ref_python.setup()
from urh import colormaps  # noqa: E402
from urh.signalprocessing.Spectrogram import Spectrogram  # noqa: E402


def main():
    rng = np.random.default_rng(20260926)
    fsk = np.load(os.path.join(HERE, "fsk.npz"))["iq"]
    fsk = np.ascontiguousarray(fsk[:12_000], dtype=np.float32).view(np.complex64).reshape(-1)
    tone = (np.exp(2j * np.pi * 0.12 * np.arange(5000)) + 0.25 * np.exp(-2j * np.pi * 0.31 * np.arange(5000))
            + 0.01 * (rng.standard_normal(5000) + 1j * rng.standard_normal(5000))).astype(np.complex64)
    short = tone[:300].copy()                                   # shorter than one window: zero padded
    silent = np.zeros(3000, dtype=np.complex64); silent[1000:1200] = tone[:200]      # exact zeros: -inf dB
    colormap = np.ascontiguousarray(colormaps.chosen_colormap_numpy_bgra, dtype=np.uint8)
    out = {"colormap": colormap}
    names = []
    for name, x, ws, ov in (("fsk_1024", fsk, 1024, 0.5), ("tone_256", tone, 256, 0.5), ("tone_512_ov75", tone, 512, 0.75),
                            ("short_1024", short, 1024, 0.5), ("silent_128", silent, 128, 0.0)):
        sp = Spectrogram(x, window_size=ws, overlap_factor=ov)
        st = sp.stft(x)
        db = sp._Spectrogram__calculate_spectrogram(x)
        with np.errstate(all="ignore"):
            img = Spectrogram.apply_bgra_lookup(db, colormap, sp.data_min, sp.data_max)
        out[name + "_x"] = x
        out[name + "_args"] = np.array([ws, ov], dtype=np.float64)
        out[name + "_stft"] = st
        out[name + "_db"] = db
        out[name + "_img"] = img
        names.append(name)
    out["names"] = np.array(names)
    os.makedirs(os.path.join(HERE, "spectrogram"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "spectrogram", "spectrogram.npz"), **out)
    print("wrote", names)


if __name__ == "__main__":
    main()
