"""The reference-side hook of INTEGRATION.md section 1: rebinds the thirteen functions URH's Python imports from urh.cythonext.signal_functions /
auto_interpretation / util (Signal.py:11, ProtocolAnalyzer.py:10, Filter.py:7, AutoInterpretation.py:8-10, Wavelet.py:3, IQArray.py:8) to the
ctypes mirrors over liburhgpu.so -- IF this host has a usable GPU.

    # src/urh/cythonext/__init__.py (or a site hook), before Signal / AutoInterpretation are imported
    import os
    if os.environ.get("URH_GPU"):
        from urh_amd import urh_hook
        urh_hook.install()

Without the library or without a GPU (urhgpu_ctx_create -> URHGPU_ERR_NO_DEVICE) the hook says so through URH's logger and leaves the Cython
names bound ("fail loudly and fall back to the Cython path if no GPU").  That is URH keeping its own
functions, decided once at start-up on the reference's side of the boundary -- the library itself has no CPU path: every urh_amd function
raises without a GPU (tests/test_abi.py::test_no_cpu_fallback)."""
import importlib

BIND = {   # module of urh.cythonext -> names rebound (same signatures, return types, edge cases and exceptions)
    "signal_functions": ("afp_demod", "grab_pulse_lens", "get_center_thresholds", "fir_filter", "iir_filter", "modulate_c"),
    "auto_interpretation": ("segment_messages_from_magnitudes", "get_threshold_divisor_histogram", "merge_plateaus", "get_plateau_lengths",
                            "median_filter"),
    "util": ("minmax", "get_magnitudes"),
}


def probe():
    """(usable, reason): creates the calling thread's default context (urhgpu_ctx_create), i.e. loads the library and finds the GPU."""
    try:
        from urh_amd import _lib
        _lib.default_context()
        return True, "ok"
    except Exception as e:               # noqa: BLE001  (UrhGpuError ERR_NO_DEVICE, a missing library, a loader error: all mean "not here")
        return False, f"{type(e).__name__}: {e}"


def install(logger=None, wrap=None):
    """Rebind the names when the GPU is usable; returns (installed, reason).  logger: an object with .warning / .info (default: URH's own,
    urh.util.Logger.logger); wrap(key, fn) -> fn: lets a caller count calls (tests/dropin_driver.py)."""
    if logger is None:
        try:
            from urh.util.Logger import logger
        except Exception:                # noqa: BLE001
            import logging
            logger = logging.getLogger("urh")
    usable, reason = probe()
    if not usable:
        logger.warning("URH_GPU: liburhgpu.so cannot be used on this host (%s): keeping the Cython functions", reason)
        return False, reason
    for module, names in BIND.items():
        gpu, cy = importlib.import_module("urh_amd." + module), importlib.import_module("urh.cythonext." + module)
        for name in names:
            fn = getattr(gpu, name)
            setattr(cy, name, wrap(module + "." + name, fn) if wrap else fn)
    logger.info("URH_GPU: demodulation / digitization functions bound to liburhgpu.so")
    return True, "ok"
