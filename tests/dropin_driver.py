#!/usr/bin/env python3
"""Driver of tests/test_reference_dropin.py (run as a subprocess so that the reference's `tests` package and `urh`
package are imported into a clean interpreter).

Applies INTEGRATION.md §1's function-level drop-in -- the names urh.cythonext.signal_functions exports are rebound to
urh_amd.signal_functions (ctypes -> liburhgpu.so) BEFORE the reference's Signal / ProtocolAnalyzer are imported -- and then
runs the reference's OWN tests/test_demodulations.py (staged by oracle/build_ref.py into the git-ignored
oracle/_ref/reftests, together with the captures it reads) with unittest.  Every patched function is wrapped in a call
counter so that the caller can see that the GPU library, not the Cython module, did the work.
Prints one JSON line: {"ran", "failures", "errors", "skipped", "calls": {...}, "details": [...]}.
"""
import json
import os
import sys
import unittest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PATCHED = ("afp_demod", "grab_pulse_lens", "get_center_thresholds", "fir_filter", "iir_filter", "modulate_c")


def main():
    import build_ref
    import ref_python
    if not ref_python.available() or not build_ref.staged():
        print(json.dumps({"unavailable": True}))
        return 0
    ref_python.setup()
    sys.path.insert(0, build_ref.REFTESTS)            # the reference's `tests` package (a regular package: wins over ours)

    # --- INTEGRATION.md §1 ------------------------------------------------------------------------------------------
    import urh_amd.signal_functions as gpu_sf
    import urh.cythonext.signal_functions as cy_sf
    calls = {name: 0 for name in PATCHED}

    def counted(name, fn):
        def wrapper(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        wrapper.__name__ = name
        return wrapper

    use_gpu = "--no-patch" not in sys.argv           # --no-patch: BASELINE configs[0], the Cython path itself (CPU plumbing run)
    for name in PATCHED:
        setattr(cy_sf, name, counted(name, getattr(gpu_sf if use_gpu else cy_sf, name)))
    # -------------------------------------------------------------------------------------------------------------------

    import tests.test_demodulations as ref_tests
    assert os.path.realpath(ref_tests.__file__).startswith(os.path.realpath(build_ref.REFTESTS)), ref_tests.__file__
    suite = unittest.defaultTestLoader.loadTestsFromModule(ref_tests)
    res = unittest.TextTestRunner(stream=sys.stderr, verbosity=2).run(suite)
    out = {"ran": res.testsRun, "failures": len(res.failures), "errors": len(res.errors), "skipped": len(res.skipped),
           "calls": calls, "details": [str(t) + "\n" + tb for t, tb in res.failures + res.errors]}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
