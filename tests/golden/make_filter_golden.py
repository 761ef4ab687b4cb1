#!/usr/bin/env python3
"""Run the REAL reference's signal_functions.fir_filter / iir_filter (oracle/_ref, this container only) on seeded inputs
and store inputs + outputs in tests/golden/filter/fir_iir.npz (the reference has no asserting test for iir_filter; these
vectors are what pins it).

    python tests/golden/make_filter_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

# name, samples, feed-forward coefficients, feedback coefficients
IIR_CASES = [("iir_1_0", 50, 1, 0), ("iir_3_2", 3000, 3, 2), ("iir_2_4", 2000, 2, 4), ("iir_5_5", 4097, 5, 5), ("iir_short", 4, 3, 2)]
FIR_CASES = [("fir_1", 100, 1), ("fir_8", 1000, 8), ("fir_64", 3000, 64), ("fir_longer_than_signal", 20, 33)]


def main():
    sf, _, _ = build_ref.import_ref()
    rng = np.random.default_rng(20260925)
    out, names = {}, []
    for name, n, na, nb in IIR_CASES:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        a, b = rng.standard_normal(na) * 0.3, rng.standard_normal(nb) * 0.3
        out[name + "_x"], out[name + "_a"], out[name + "_b"] = x, a, b
        out[name + "_y"] = np.asarray(sf.iir_filter(a, b, x))
        names.append(name)
    for name, n, m in FIR_CASES:
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        h = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
        out[name + "_x"], out[name + "_h"] = x, h
        out[name + "_y"] = np.asarray(sf.fir_filter(x, h))
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "filter", "fir_iir.npz"), **out)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    main()
