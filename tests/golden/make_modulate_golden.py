#!/usr/bin/env python3
"""Run the REAL reference's signal_functions.modulate_c (oracle/_ref, this container only) on seeded bit strings and
store arguments + results in tests/golden/modulate/modulate.npz.  The generator's sinf / cosf come from the host's
glibc (FMA build on every x86-64 host with FMA + AVX2), so the vectors are valid for hosts of that kind -- the same
caveat as for the atan2f-dependent goldens.

    python tests/golden/make_modulate_golden.py
"""
import array
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

CASES = [
    # name, mod, parameters, bits_per_symbol, sps, n_bits, pause, start, dtype, amplitude
    ("fsk2_f32", "FSK", [-20e3, 20e3], 1, 100, 300, 76, 0, "float32", 1.0),
    ("fsk2_late", "FSK", [-20e3, 20e3], 1, 100, 200, 0, 7_340_032, "float32", 1.0),
    ("fsk4_i16", "FSK", [-30e3, -10e3, 10e3, 30e3], 2, 40, 400, 33, 1234, "int16", 32767.0),
    ("ask2_f32", "ASK", [0.0, 1.0], 1, 50, 200, 500, 0, "float32", 1.0),
    ("ask4_i8", "ASK", [0.0, 42.0, 85.0, 127.0], 2, 25, 300, 0, 999, "int8", 127.0),
    ("psk2_f32", "PSK", [-np.pi / 2, np.pi / 2], 1, 8, 96, 10, 0, "float32", 1.0),
    ("psk4_f32", "PSK", list(np.array([-135.0, -45.0, 45.0, 135.0]) * np.pi / 180), 2, 100, 200, 76, 50_000, "float32", 1.0),
    ("oqpsk_f32", "OQPSK", list(np.array([-135.0, -45.0, 45.0, 135.0]) * np.pi / 180), 2, 100, 200, 76, 0, "float32", 1.0),
    ("oqpsk_i8_odd", "OQPSK", list(np.array([-135.0, -45.0, 45.0, 135.0]) * np.pi / 180), 2, 10, 31, 5, 777, "int8", 100.0),
]


def main():
    sf, _, _ = build_ref.import_ref()
    rng = np.random.default_rng(20260924)
    out = {}
    names = []
    for name, mod, par, bps, sps, nb, pause, start, dtype, amp in CASES:
        bits = rng.integers(0, 2, nb).astype(np.uint8)
        r = sf.modulate_c(array.array("B", bits.tolist()), sps, mod, array.array("f", par), bps, amp, 40e3, 0.25, 1e6, pause, start,
                          np.dtype(dtype).type)
        out[name + "_bits"] = bits
        out[name + "_par"] = np.array(par, dtype=np.float32)
        out[name + "_args"] = np.array([bps, sps, pause, start], dtype=np.int64)
        out[name + "_amp"] = np.array([amp, 40e3, 0.25, 1e6], dtype=np.float32)
        out[name + "_out"] = r
        names.append(f"{name}:{mod}:{dtype}")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "modulate", "modulate.npz"), **out)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    main()
