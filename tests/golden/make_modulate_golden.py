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


GFSK_CASES = [
    # name, parameters, bits_per_symbol, sps, n_bits, pause, start, dtype, amplitude, gauss_bt, filter_width, sample_rate
    ("gfsk_ref_test", [-10e3, 10e3], 1, 100, 10, 1000, 0, "float32", 1.0, 0.5, 1.0, 1e6),      # test_modulator.py:154-157
    ("gfsk_late", [-20e3, 20e3], 1, 40, 300, 7, 123_456, "float32", 1.0, 0.3, 2.0, 2e6),
    ("gfsk4_i16", [-30e3, -10e3, 10e3, 30e3], 2, 25, 400, 33, 16_777_210, "int16", 32767.0, 0.5, 1.0, 1e6),
    ("gfsk_short", [-20e3, 20e3], 1, 8, 3, 5, 99, "int8", 127.0, 0.5, 4.0, 1e6),               # filter longer than the message
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
    # GFSK: the Gaussian-filtered frequencies come out of numpy's float32 BLAS convolution (host dependent in the last bit);
    # the vectors carry numpy's frequencies of THIS host next to the reference's output, so that everything downstream of the
    # convolution (phase recurrence, carrier) can be checked bit for bit anywhere.
    import urh_oracle
    gnames = []
    for name, par, bps, sps, nb, pause, start, dtype, amp, bt, fw, rate in GFSK_CASES:
        bits = rng.integers(0, 2, nb).astype(np.uint8)
        r = sf.modulate_c(array.array("B", bits.tolist()), sps, "GFSK", array.array("f", par), bps, amp, 40e3, 0.25, rate, pause, start,
                          np.dtype(dtype).type, bt, fw)
        gfir = urh_oracle.gauss_fir(rate, sps, bt, fw)
        sym = bits[:nb // bps * bps].reshape(-1, bps)
        index = (sym.astype(np.int64) << np.arange(bps - 1, -1, -1)).sum(axis=1)
        raw = np.repeat(np.array(par, np.float32)[index], sps).astype(np.float32)
        freqs = np.convolve(raw, gfir, mode="same") if len(raw) >= len(gfir) else np.convolve(gfir, raw, mode="same")[:len(raw)]
        out[name + "_bits"], out[name + "_par"], out[name + "_freqs"], out[name + "_gfir"] = bits, np.array(par, np.float32), freqs, gfir
        out[name + "_args"] = np.array([bps, sps, pause, start], dtype=np.int64)
        out[name + "_amp"] = np.array([amp, 0.25, rate, bt, fw], dtype=np.float32)
        out[name + "_out"] = r
        gnames.append(f"{name}:{dtype}")
    out["gfsk_names"] = np.array(gnames)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "modulate", "modulate.npz"), **out)
    print("wrote", len(names), "cases")


if __name__ == "__main__":
    main()
