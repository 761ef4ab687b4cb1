#!/usr/bin/env python3
"""Driver of tests/test_reference_dropin.py (run as a subprocess so that the reference's `tests` package and `urh`
package are imported into a clean interpreter).

Applies INTEGRATION.md §1's function-level drop-in -- the names the reference's Python imports from urh.cythonext.signal_functions,
urh.cythonext.auto_interpretation and urh.cythonext.util are rebound to urh_amd.signal_functions / auto_interpretation / util
(ctypes -> liburhgpu.so) BEFORE the reference's Signal / ProtocolAnalyzer / AutoInterpretation are imported -- and then runs the
reference's OWN headless hot-path tests (staged by oracle/build_ref.py into the git-ignored oracle/_ref/reftests, together with the
captures they read) with unittest:

    tests/test_demodulations.py, tests/test_protocol_analyzer.py, tests/test_iq_array.py, tests/test_modulator.py,
    tests/auto_interpretation/test_*.py

plus the known-answer test of tests/test_filter.py:20-31 (its class needs the GUI form in setUp; the same assertion is made here on
the reference's Filter object).  Every patched function is wrapped in a call counter so that the caller can see that the GPU
library, not the Cython module, did the work.  `--no-patch`: the Cython path itself (BASELINE configs[0], the CPU plumbing run).
`--hook`: the binding goes through urh_amd.urh_hook.install() -- on a host without a GPU it declines (logged) and the reference's tests run
on URH's own Cython functions (tests/test_oracle.py::test_hook_keeps_cython_without_gpu).
Prints one JSON line: {"ran", "failures", "errors", "skipped", "per_module": {...}, "calls": {...}, "details": [...]}.
"""
import json
import os
import sys
import unittest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PATCHED = {
    "signal_functions": ("afp_demod", "grab_pulse_lens", "get_center_thresholds", "fir_filter", "iir_filter", "modulate_c"),
    "auto_interpretation": ("segment_messages_from_magnitudes", "get_threshold_divisor_histogram", "merge_plateaus", "get_plateau_lengths",
                            "median_filter"),
    "util": ("minmax", "get_magnitudes"),
}
MODULES = ["tests.test_demodulations", "tests.test_protocol_analyzer", "tests.test_iq_array", "tests.test_modulator",
           "tests.auto_interpretation.test_additional_signals", "tests.auto_interpretation.test_auto_interpretation_integration",
           "tests.auto_interpretation.test_bit_length_detection", "tests.auto_interpretation.test_center_detection",
           "tests.auto_interpretation.test_estimate_tolerance", "tests.auto_interpretation.test_message_segmentation",
           "tests.auto_interpretation.test_modulation_detection", "tests.auto_interpretation.test_noise_detection"]


class FirFilterKnownAnswer(unittest.TestCase):
    """tests/test_filter.py:20-31 (TestFilter.test_fir_filter) on the reference's own Filter object"""

    def test_fir_filter(self):
        import numpy as np
        from urh.signalprocessing.Filter import Filter
        input_signal = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 42], dtype=np.complex64)
        filtered = Filter([0.25, 0.25, 0.25, 0.25]).apply_fir_filter(input_signal.flatten())
        expected = np.array([0.25, 0.75, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 16.5], dtype=np.complex64)
        self.assertTrue(np.array_equal(filtered, expected))


def main():
    import build_ref
    import ref_python
    if not ref_python.available() or not build_ref.staged():
        print(json.dumps({"unavailable": True}))
        return 0
    ref_python.setup()
    sys.path.insert(0, build_ref.REFTESTS)            # the reference's `tests` package (a regular package: wins over ours)

    # --- INTEGRATION.md §1 ------------------------------------------------------------------------------------------
    import importlib
    calls = {}

    def counted(key, fn):
        def wrapper(*a, **k):
            calls[key] += 1
            return fn(*a, **k)
        wrapper.__name__ = key.split(".")[-1]
        return wrapper

    use_gpu = "--no-patch" not in sys.argv           # --no-patch: BASELINE configs[0], the Cython path itself (CPU plumbing run)
    hook = None
    if "--hook" in sys.argv:
        # the hook a maintainer installs (urh_amd/urh_hook.py): binds the GPU functions when a GPU is usable,
        # says why not and keeps URH's Cython functions otherwise.  The counters then tell which side did the work.
        from urh_amd import urh_hook
        assert urh_hook.BIND == PATCHED
        for mod, names in PATCHED.items():
            for name in names:
                calls[mod + "." + name] = 0
        log = []

        class _Log:
            def warning(self, fmt, *a): log.append("warning: " + fmt % a)
            def info(self, fmt, *a): log.append("info: " + fmt % a)
        installed, reason = urh_hook.install(logger=_Log(), wrap=counted)
        hook = {"installed": installed, "reason": reason, "log": log}
    else:
        for mod, names in PATCHED.items():
            cy = importlib.import_module("urh.cythonext." + mod)
            gpu = importlib.import_module("urh_amd." + mod)
            for name in names:
                key = mod + "." + name
                calls[key] = 0
                setattr(cy, name, counted(key, getattr(gpu if use_gpu else cy, name)))
    # -------------------------------------------------------------------------------------------------------------------

    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    per_module, details = {}, []
    total = {"ran": 0, "failures": 0, "errors": 0, "skipped": 0}
    loader = unittest.defaultTestLoader
    suites = []
    for name in MODULES:
        if only and not any(o in name for o in only):
            continue
        mod = importlib.import_module(name)
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(build_ref.REFTESTS)), mod.__file__
        suites.append((name, loader.loadTestsFromModule(mod)))
    if not only or any("filter" in o for o in only):
        suites.append(("tests.test_filter:20-31", loader.loadTestsFromTestCase(FirFilterKnownAnswer)))
    for name, suite in suites:
        res = unittest.TextTestRunner(stream=sys.stderr, verbosity=1).run(suite)
        rec = {"ran": res.testsRun, "failures": len(res.failures), "errors": len(res.errors), "skipped": len(res.skipped)}
        per_module[name] = rec
        for k in total:
            total[k] += rec[k]
        details += [name + " :: " + str(t) + "\n" + tb for t, tb in res.failures + res.errors]
    out = dict(total, per_module=per_module, calls=calls, details=details, hook=hook)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
