#!/usr/bin/env python3
"""Run the REAL reference's AutoInterpretation.estimate (this container only) on the IQ of every float32 golden capture
(all dtypes) and store what it returns in tests/golden/estimates.json; the GPU test compares urh_amd.estimators.estimate_dev with it
(`detect`: modulation=None, i.e. detect_modulation_for_messages decides).

    python tests/golden/make_estimate_golden.py
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
from urh.ainterpretation import AutoInterpretation as AI  # noqa: E402

out = {}
for f in sorted(os.listdir(HERE)):
    if not f.endswith(".npz"):
        continue
    z = np.load(os.path.join(HERE, f))
    iq = z["iq"]
    mod = str(z["modulation_type"])
    for m in ([mod, "OOK", None] if mod == "ASK" else [mod, None]):
        for noise in (None, float(z["noise_threshold"])):
            r = AI.estimate(iq, noise=noise, modulation=m)
            key = f"{f[:-4]}|{m if m is not None else 'detect'}|{'auto' if noise is None else 'given'}"
            out[key] = None if r is None else {k: (float(v) if k in ("center", "noise") else (int(v) if k != "modulation_type" else v))
                                               for k, v in r.items()}
            print(key, out[key])
json.dump(out, open(os.path.join(HERE, "estimates.json"), "w"), indent=1, sort_keys=True)
