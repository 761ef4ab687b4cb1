#!/usr/bin/env python3
"""Run the REAL reference's path_creator.create_path (oracle/_ref build + the PyQt6 stub, whose QByteArray /
QDataStream / QPainterPath keep the serialised vertices; this container only) and store arguments and the decoded
vertex arrays in tests/golden/path/paths.npz.

    python tests/golden/make_path_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_python  # noqa: E402

ref_python.setup()
from urh.cythonext import path_creator  # noqa: E402


def main():
    rng = np.random.default_rng(20260925)
    qad = np.load(os.path.join(HERE, "fsk.npz"))["qad"].astype(np.float32)
    big = (np.sin(np.arange(300_000) * 0.0137) * (1 + 0.3 * rng.standard_normal(300_000))).astype(np.float32)
    big[1000:1020] = np.nan
    i16 = (rng.integers(-32768, 32767, 123_457)).astype(np.int16)
    u8 = (rng.integers(0, 256, 50_021)).astype(np.uint8)
    cases = [
        ("qad_full", qad, 0, len(qad), None),
        ("qad_window", qad, 1234, 60_001, [(1234, 30_000), (30_000, 45_000.5), (45_000, 60_001)]),
        ("qad_small", qad, 100, 4_000, None),                       # at most one sample per pixel: the samples themselves
        ("big_nan", big, 7, 299_990, [(7, 100_000), (100_000, 299_990)]),
        ("i16", i16, 0, len(i16), None),
        ("u8", u8, 21, 50_021, [(21, 25_000), (20_000, 50_021)]),
    ]
    out = {"names": np.array([c[0] for c in cases])}
    for name, arr, start, end, ranges in cases:
        paths = path_creator.create_path(arr, start, end, ranges)
        out[name + "_samples"] = arr
        out[name + "_args"] = np.array([start, end], dtype=np.int64)
        out[name + "_ranges"] = np.array(ranges if ranges is not None else [(start, end)], dtype=np.float64)
        out[name + "_default_ranges"] = np.array([ranges is None])
        for k, p in enumerate(paths):
            x, y = p.vertices()
            out[f"{name}_x{k}"] = x
            out[f"{name}_y{k}"] = y
    os.makedirs(os.path.join(HERE, "path"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "path", "paths.npz"), **out)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
