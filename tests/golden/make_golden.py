#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (its own Python classes on top of the
oracle/_ref Cython build) in this container.  /root/reference does not exist on the GPU box, so the
vectors are committed; re-run this script only when cases are added.

    python tests/golden/make_golden.py

Each case stores the input IQ bytes, the demodulation parameters and what the reference produced:
qad (Signal.qad), ppseq (grab_pulse_lens), flat bits / pauses / bit_sample_pos (_ppseq_to_bits), the
auto-detected noise threshold where the reference test relies on it, and -- where the reference's
own test asserts one -- the known-answer bit string (`kat`, with `kat_mode` exact|prefix).
"""
import array
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_python  # noqa: E402

ref_python.setup()
from urh.cythonext import signal_functions as sf  # noqa: E402
from urh.signalprocessing.IQArray import IQArray  # noqa: E402
from urh.signalprocessing.ProtocolAnalyzer import ProtocolAnalyzer  # noqa: E402
from urh.signalprocessing.Signal import Signal  # noqa: E402

DATA = "/root/reference/tests/data/"


def flat(lists, dtype):
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l)
    data = np.zeros(off[-1], dtype=dtype)
    for i, l in enumerate(lists):
        data[off[i]:off[i + 1]] = np.asarray(l, dtype=dtype)
    return data, off


def run_case(name, signal: Signal, kat=None, kat_mode="exact", note=""):
    pa = ProtocolAnalyzer(signal)
    qad = np.array(signal.qad, dtype=np.float32)
    ppseq = np.asarray(sf.grab_pulse_lens(signal.qad, signal.center, signal.tolerance, signal.modulation_type,
                                          signal.samples_per_symbol, signal.bits_per_symbol, signal.center_spacing),
                       dtype=np.int64).reshape(-1, 2)
    bit_data, pauses, pos = pa._ppseq_to_bits(ppseq, signal.samples_per_symbol, signal.bits_per_symbol,
                                              pause_threshold=signal.pause_threshold)
    pa.get_protocol_from_signal()
    bits_str = pa.plain_bits_str
    if kat is not None:
        if kat_mode == "exact":
            assert bits_str[0] == kat, (name, bits_str[0], kat)
        else:
            assert bits_str[0].startswith(kat), (name, bits_str[0], kat)
    bits, msg_off = flat(bit_data, np.uint8)
    posf, pos_off = flat(pos, np.int64)
    out = dict(
        iq=np.ascontiguousarray(signal.iq_array.data),
        modulation_type=signal.modulation_type, bits_per_symbol=signal.bits_per_symbol,
        noise_threshold=np.float64(signal.noise_threshold), center=np.float64(signal.center),
        center_spacing=np.float64(signal.center_spacing), tolerance=signal.tolerance,
        samples_per_symbol=signal.samples_per_symbol, pause_threshold=signal.pause_threshold,
        costas_loop_bandwidth=np.float64(signal.costas_loop_bandwidth),
        qad=qad, ppseq=ppseq, bits=bits, msg_off=msg_off, pauses=np.asarray(pauses, dtype=np.int64),
        pos=posf, pos_off=pos_off, kat=kat or "", kat_mode=kat_mode, note=note,
    )
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: n={len(qad)} rows={len(ppseq)} msgs={len(pauses)} bits={len(bits)} noise={signal.noise_threshold}")


def main():
    # ---- /root/reference/tests/test_demodulations.py -------------------------------------------------
    s = Signal(DATA + "ask.complex", "ASK-Test")                       # :14-27
    s.modulation_type = "ASK"; s.samples_per_symbol = 295; s.center = 0.0219
    run_case("ask", s, "1011001001011011011011011011011011001000000", "prefix", "test_demodulations.py:14-27")

    s = Signal(DATA + "ask_short.complex", "ASK-Test2")                # :29-40
    s.modulation_type = "ASK"; s.noise_threshold = 0.0299; s.samples_per_symbol = 16; s.center = 0.1300; s.tolerance = 0
    run_case("ask_short", s, "10101010", "exact", "test_demodulations.py:29-40")

    s = Signal(DATA + "fsk.complex", "FSK-Test")                       # :42-53
    s.modulation_type = "FSK"; s.samples_per_symbol = 100; s.center = 0
    run_case("fsk", s,
             "101010101010101010101010101010101100011000100110110001100010011011110100110111000001110110011000111011101111011110100100001001111001100110011100110100100011100111010011111100011",
             "exact", "test_demodulations.py:42-53")

    bits = array.array("B", list(map(int, "101010")))                  # :55-72
    res = sf.modulate_c(bits, 8, "FSK", array.array("f", [-10e3, 10e3]), 1, 1, 40e3, 0, 1e6, 1000, 0)
    s = Signal(""); s.iq_array = IQArray(res); s.qad_center = 0; s.samples_per_symbol = 8
    run_case("fsk_sps8", s, "101010", "exact", "test_demodulations.py:55-72")

    s = Signal(DATA + "psk_gen_noisy.complex", "PSK-Test")             # :74-87
    s.modulation_type = "PSK"; s.samples_per_symbol = 300; s.center = 0; s.noise_threshold = 0; s.tolerance = 10
    run_case("psk", s, "1011", "prefix", "test_demodulations.py:74-87")

    bits = array.array("B", [1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 0, 1, 0, 1])      # :89-120
    params = array.array("f", [np.pi * a / 180 for a in [-135, -45, 45, 135]])
    res = sf.modulate_c(bits, 100, "PSK", params, 2, 1, 40e3, 0, 1e6, 1000, 0)
    s = Signal(""); s.iq_array = IQArray(res); s.bits_per_symbol = 2; s.center = 0; s.center_spacing = 1
    s.modulation_type = "PSK"
    run_case("psk4_clean", s, "10101010", "prefix", "test_demodulations.py:89-107")
    np.random.seed(42)
    noised = res + 0.1 * np.random.normal(loc=0, scale=np.sqrt(2) / 2, size=(len(res), 2))
    s = Signal(""); s.iq_array = IQArray(noised.astype(np.float32)); s.bits_per_symbol = 2; s.center = 0
    s.modulation_type = "PSK"; s.center_spacing = 1.5; s.noise_threshold = 0.2
    run_case("psk4_noisy", s, "10101010", "prefix", "test_demodulations.py:108-120")

    bits = array.array("B", [1, 0, 1, 0, 1, 1, 0, 0, 0, 1])             # :122-135
    res = sf.modulate_c(bits, 100, "FSK", array.array("f", [-20e3, -10e3, 10e3, 20e3]), 2, 1, 40e3, 0, 1e6, 1000, 0)
    s = Signal(""); s.iq_array = IQArray(res); s.bits_per_symbol = 2; s.center = 0; s.center_spacing = 0.1
    run_case("fsk4", s, "1010110001", "exact", "test_demodulations.py:122-135")

    # ---- /root/reference/tests/test_protocol_analyzer.py:25-39 ----------------------------------------
    s = Signal(DATA + "steckdose_anlernen.complex", "RWE")
    s.noise_threshold = 0.06; s.center = 0; s.samples_per_symbol = 100; s.pause_threshold = 8
    run_case("steckdose", s,
             "101010101010101010101010101010101001101001111101100110100111110111010010011000010110110101111"
             "010111011011000011000101000010001001101100101111010110100110011100100110000101001110100001111"
             "111101000111001110000101110100100111010110110100001101101101010100011011010001010110011100011"
             "010100010101111110011010011001000000110010011010001000100100100111101110110010011111011100010"
             "10110010100011111101110111000010111100111101001011101101011011010110101011100",
             "exact", "test_protocol_analyzer.py:25-39")

    # ---- integer dtypes (tests/test_protocol_analyzer.py:46-61 uses two_participants.complex16s) -----
    s = Signal(DATA + "two_participants.complex16s", "2p")
    s.noise_threshold = 0; s.center = -0.0507; s.samples_per_symbol = 100; s.tolerance = 5   # centre as in the test
    s.iq_array = IQArray(s.iq_array.data[:400000].copy())
    run_case("two_participants_i8", s, None, note="test_protocol_analyzer.py:46-61 (first 400k samples, int8)")

    s = Signal(DATA + "homematic.complex32s", "hm")
    s.modulation_type = "FSK"; s.samples_per_symbol = 100; s.center = 0.0
    run_case("homematic_i16", s, None, note="auto noise threshold, int16 capture")

    s = Signal(DATA + "enocean.complex", "eno")
    s.modulation_type = "ASK"; s.samples_per_symbol = 40; s.center = 0.0975; s.tolerance = 1; s.noise_threshold = 0.0077
    run_case("enocean_ask", s, None, note="params close to test_auto_interpretation_integration.py:62-90")


if __name__ == "__main__":
    main()
