#!/usr/bin/env python3
"""Run the REAL reference's ProtocolAnalyzer.get_protocol_from_signal (this container only) on every golden capture with
the parameters stored in its .npz and record what ends up in the Message objects: bits, pause, RSSI, timestamp,
bit_sample_pos -- also with message_length_divisor = 8 for the ASK captures.  -> tests/golden/messages.json

PSK captures: the reference's Costas loop never writes result[0] (np.empty: uninitialised memory, signal_functions.pyx:264, :288), so
what its first sample is classified as is not reproducible.  For these captures numpy.empty is patched to hand out float32 arrays
filled with -4.0 (NOISE_FSK_PSK) -- the value this library documents for that element (include/urhgpu.h) -- which pins the
reference's result[0] without touching anything it computes.

    python tests/golden/make_messages_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_python  # noqa: E402

ref_python.setup()
from urh.signalprocessing.IQArray import IQArray  # noqa: E402
from urh.signalprocessing.ProtocolAnalyzer import ProtocolAnalyzer  # noqa: E402
from urh.signalprocessing.Signal import Signal  # noqa: E402

out = {}
for f in sorted(os.listdir(HERE)):
    if not f.endswith(".npz"):
        continue
    z = np.load(os.path.join(HERE, f))
    mod = str(z["modulation_type"])
    _empty = np.empty
    if mod == "PSK":
        def _filled(shape, dtype=float, *a, **k):          # see the docstring: result[0] of the Costas loop
            arr = _empty(shape, dtype, *a, **k)
            if np.dtype(dtype) == np.float32:
                arr.fill(-4.0)
            return arr
        np.empty = _filled
    for divisor in ([1, 8] if mod == "ASK" else [1]):
        s = Signal("")
        s.iq_array = IQArray(z["iq"])
        s.modulation_type = mod
        s.bits_per_symbol = int(z["bits_per_symbol"])
        s.noise_threshold = float(z["noise_threshold"])
        s.center = float(z["center"])
        s.center_spacing = float(z["center_spacing"])
        s.tolerance = int(z["tolerance"])
        s.samples_per_symbol = int(z["samples_per_symbol"])
        s.pause_threshold = int(z["pause_threshold"])
        s.costas_loop_bandwidth = float(z["costas_loop_bandwidth"])
        s.message_length_divisor = divisor
        pa = ProtocolAnalyzer(s)
        pa.get_protocol_from_signal()
        out[f"{f[:-4]}|{divisor}"] = [dict(bits=m.plain_bits_str, pause=int(m.pause), rssi=float(m.rssi), timestamp=float(m.timestamp),
                                           pos=[int(v) for v in m.bit_sample_pos]) for m in pa.messages]
        print(f, divisor, len(pa.messages), [round(m.rssi, 4) for m in pa.messages[:3]])
    np.empty = _empty
json.dump(out, open(os.path.join(HERE, "messages.json"), "w"))
