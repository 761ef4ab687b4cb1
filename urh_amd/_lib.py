"""ctypes binding of liburhgpu.so (the C ABI declared in include/urhgpu.h).

The library is the product: there is NO CPU fallback.  Importing this module only loads the shared
object (that works without a GPU, e.g. for the symbol-export test); creating a context on a
machine without a GPU raises `UrhGpuError`.
"""
import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liburhgpu.so")

OK = 0
ERR_HIP, ERR_DTYPE, ERR_ARG, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6
BLOB_LEN16 = 2            # header[7] bit 1 of a compact blob: 16-bit row lengths + escape list (include/urhgpu.h)
BLOB_ROW16 = 4       # header[7] bit 2: state and length of a row in one uint16 (include/urhgpu.h)

DT_I8, DT_U8, DT_I16, DT_U16, DT_F32 = 0, 1, 2, 3, 4
MOD_ASK, MOD_FSK, MOD_PSK, MOD_OTHER = 0, 1, 2, 3
ROW_ABSORBED = -(1 << 62)
BLOB_MAGIC = 0x55524842424C4F42           # URHGPU_BLOB_MAGIC
BLOB_HEADER_BYTES = 128


class UrhGpuError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class Params(C.Structure):
    """struct urhgpu_params (include/urhgpu.h)"""
    _fields_ = [
        ("dtype", C.c_int), ("mod", C.c_int), ("bits_per_symbol", C.c_int),
        ("noise_threshold", C.c_float), ("center", C.c_float), ("center_spacing", C.c_float),
        ("tolerance", C.c_int), ("samples_per_symbol", C.c_uint32), ("costas_loop_bandwidth", C.c_float),
        ("pause_threshold", C.c_int64), ("write_bit_sample_pos", C.c_int), ("noise_other", C.c_float),
        ("mod_order", C.c_int),
    ]


class Outputs(C.Structure):
    """struct urhgpu_outputs (include/urhgpu.h); all pointers are device pointers"""
    _fields_ = [
        ("qad", C.c_void_p), ("rows", C.c_void_p), ("cap_rows", C.c_int64),
        ("bits", C.c_void_p), ("cap_bits", C.c_int64),
        ("msg_off", C.c_void_p), ("pauses", C.c_void_p), ("cap_msg", C.c_int64),
        ("pos", C.c_void_p), ("cap_pos", C.c_int64), ("pos_off", C.c_void_p),
        ("counts", C.c_void_p),
        ("blob", C.c_void_p), ("cap_blob", C.c_int64), ("h_counts", C.c_void_p),
    ]


class HostResult(C.Structure):
    """struct urhgpu_host_result (include/urhgpu.h); pointers are pinned host memory owned by the stream"""
    _fields_ = [
        ("seq", C.c_int64), ("n_samples", C.c_int64),
        ("n_rows", C.c_int64), ("n_msg", C.c_int64), ("n_bits", C.c_int64), ("n_pos", C.c_int64), ("rows_needed", C.c_int64),
        ("blob_bytes", C.c_int64), ("truncated", C.c_int),
        ("row_len", C.c_void_p), ("row_len16", C.c_void_p), ("esc", C.c_void_p), ("n_esc", C.c_int64),
        ("row_state", C.c_void_p), ("bits_packed", C.c_void_p),
        ("msg_off", C.c_void_p), ("pauses", C.c_void_p), ("pos_off", C.c_void_p), ("pos32", C.c_void_p),
        ("blob", C.c_void_p), ("d_qad", C.c_void_p), ("row16", C.c_void_p),
    ]


# every symbol include/urhgpu.h declares: name -> (restype, argtypes)
_vp, _i64, _i, _f = C.c_void_p, C.c_int64, C.c_int, C.c_float
PROTOTYPES = {
    "urhgpu_version": (_i, []),
    "urhgpu_strerror": (C.c_char_p, [_i]),
    "urhgpu_last_hip_error": (C.c_char_p, []),
    "urhgpu_device_count": (_i, [C.POINTER(_i)]),
    "urhgpu_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "urhgpu_ctx_destroy": (_i, [_vp]),
    "urhgpu_ctx_set_stream": (_i, [_vp, _vp]),
    "urhgpu_ctx_use_private_stream": (_i, [_vp]),
    "urhgpu_ctx_sync": (_i, [_vp]),
    "urhgpu_ctx_set_pipelined": (_i, [_vp, _i, _vp]),
    "urhgpu_ctx_set_tuning": (_i, [_vp, C.c_char_p, _i]),
    "urhgpu_ctx_join": (_i, [_vp]),
    "urhgpu_ctx_reserve": (_i, [_vp, _i64, _i]),
    "urhgpu_ctx_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64), C.c_char_p, _i]),
    "urhgpu_ctx_costas_stats": (_i, [_vp, C.POINTER(C.c_int32)]),
    "urhgpu_ctx_profile_begin": (_i, [_vp, _i]),
    "urhgpu_ctx_profile_end": (_i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "urhgpu_get_magnitudes": (_i, [_vp, _vp, _i, _i64, _vp]),
    "urhgpu_afp_demod": (_i, [_vp, _vp, _i, _i64, _f, _i, _i, _f, _f, _vp]),
    "urhgpu_get_center_thresholds": (_i, [_f, _f, _i, _vp]),
    "urhgpu_grab_pulse_lens": (_i, [_vp, _vp, _i64, _f, C.c_uint16, _i, C.c_uint32, C.c_uint8, _f, _f, _vp, _i64,
                                    C.POINTER(_i64)]),
    "urhgpu_ppseq_to_bits": (_i, [_vp, _vp, _i64, _i64, _i, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "urhgpu_fir_filter": (_i, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "urhgpu_fir_filter_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "urhgpu_fir_filter_stats_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp]),
    "urhgpu_bandpass": (_i, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "urhgpu_bandpass_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i]),
    "urhgpu_iir_filter": (_i, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "urhgpu_afp_demod_dev": (_i, [_vp, _vp, _i64, C.POINTER(Params), _vp]),
    "urhgpu_grab_pulse_lens_dev": (_i, [_vp, _vp, _i64, C.POINTER(Params), _vp, _i64, _vp]),
    "urhgpu_ppseq_to_bits_dev": (_i, [_vp, _vp, _vp, _i64, C.POINTER(Params), C.POINTER(Outputs)]),
    "urhgpu_iq_to_bits_dev": (_i, [_vp, _vp, _i64, C.POINTER(Params), C.POINTER(Outputs)]),
    "urhgpu_blob_capacity": (_i64, [_i64, _i64, _i64, _i64, _i]),
    "urhgpu_outputs_to_host": (_i, [_vp, C.POINTER(Outputs), _i, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_host_libm_check": (_i, [C.POINTER(_i64)]),
    "urhgpu_stream_capacities": (_i, [_i64, C.POINTER(Params), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "urhgpu_stream_create": (_i, [_vp, _i64, C.POINTER(Params), _i, _i, _i64, C.POINTER(_vp)]),
    "urhgpu_stream_destroy": (_i, [_vp]),
    "urhgpu_stream_push": (_i, [_vp, _vp, _i64, C.POINTER(HostResult)]),
    "urhgpu_stream_push_upload": (_i, [_vp, _vp, _vp, _i64, C.POINTER(HostResult)]),
    "urhgpu_stream_flush": (_i, [_vp, C.POINTER(HostResult), C.POINTER(_i)]),
    "urhgpu_stream_stats": (_i, [_vp, C.POINTER(_i64)]),
    "urhgpu_stream_wide_passes": (_i, [_vp, C.POINTER(_i64)]),
    "urhgpu_shard_runs_dev": (_i, [_vp, _vp, _i64, _i64, _i64, _i, _i, _vp, C.POINTER(Params), C.POINTER(Outputs), _vp]),
    "urhgpu_shard_prelaunch_dev": (_i, [_vp, _vp, _i64, _i64, _i64, _i, _i, C.POINTER(Params), C.POINTER(Outputs)]),
    "urhgpu_shard_launch_dev": (_i, [_vp, _vp, _i64, _i64, _i64, _i, _i, _vp, C.POINTER(Params), C.POINTER(Outputs)]),
    "urhgpu_shard_rows_dev": (_i, [_vp, _vp, _vp]),
    "urhgpu_shard_bits_prepare_dev": (_i, [_vp, _vp, _vp]),
    "urhgpu_shard_bits_finish_dev": (_i, [_vp, _vp]),
    "urhgpu_magnitude_chunk_stats_dev": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _vp, _vp]),
    "urhgpu_segment_runs_dev": (_i, [_vp, _vp, _i, _i64, _f, _vp, _i64, _vp]),
    "urhgpu_message_ranges_dev": (_i, [_vp, _vp, _i, _i64, _f, _vp, _i64, C.POINTER(_i64), _vp, _i64, C.POINTER(_i64), C.POINTER(_i)]),
    "urhgpu_message_ranges_demod_dev": (_i, [_vp, _vp, _i, _i64, _f, _vp, _i64, C.POINTER(_i64), _vp, _i64, C.POINTER(_i64), C.POINTER(_i), _vp]),
    "urhgpu_compact_gt_dev": (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    "urhgpu_edges_le_dev": (_i, [_vp, _vp, _i64, _f, _vp, _i64, _vp]),
    "urhgpu_minmax_f32_dev": (_i, [_vp, _vp, _i64, _vp]),
    "urhgpu_pairwise_sum_f32_dev": (_i, [_vp, _vp, _i64, _i, _f, C.POINTER(_f)]),
    "urhgpu_histogram_f32_dev": (_i, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "urhgpu_msg_center_stats": (_i, [_vp, _vp, _i64, _vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "urhgpu_msg_plateaus": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp, _i64]),
    "urhgpu_detect_modulation_dev": (_i, [_vp, _vp, _i64, _vp, _i, _i, _i, _vp, _vp]),
    "urhgpu_msg_bit_lengths": (_i, [_vp, _vp, _i, _vp, _vp]),
    "urhgpu_msg_plateau_decisions": (_i, [_vp, _vp, _i64, _vp, _vp, _i, _i, _i64, _vp, _vp]),
    "urhgpu_msg_estimate": (_i, [_vp, _vp, _i64, _vp, _i, _i64, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "urhgpu_test_bit_length_from_counts": (_i, [_vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "urhgpu_msg_divisor_histogram": (_i, [_vp, _i64, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "urhgpu_bit_length_from_order": (_i, [_vp, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_merge_plateaus": (_i, [_vp, _i64, C.c_uint64, C.c_uint64, _vp, C.POINTER(_i64)]),
    "urhgpu_minmax": (_i, [_vp, _vp, _i, _i64, _vp]),
    "urhgpu_segment_messages": (_i, [_vp, _vp, _i, _i64, _f, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_get_plateau_lengths": (_i, [_vp, _vp, _i64, _f, _i, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_threshold_divisor_histogram": (_i, [_vp, _i64, _f, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_median_filter": (_i, [_vp, _vp, _i64, C.c_uint, _vp]),
    "urhgpu_test_hot_probe": (_i, [_vp, _vp, _i64, C.POINTER(Params), _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(_i64), _vp, _vp, _i, _i]),
    "urhgpu_test_hot_stamps": (_i, [_i]),
    "urhgpu_test_tail_skip": (_i, [_i]),
    "urhgpu_test_fetch_chunk_tables": (_i, [_vp, _vp, _i64]),
    "urhgpu_test_fast_division_dev": (_i, [_vp, C.c_uint64, _i, C.POINTER(C.c_uint64)]),
    "urhgpu_test_sincosf_fast_dev": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "urhgpu_test_force_merge_ambiguous": (_i, [_i]),
    "urhgpu_modulate_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_uint32, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_modulate": (_i, [_vp, _vp, _i64, C.c_uint32, _i, _vp, _i, _f, _f, _f, _f, C.c_uint32, C.c_uint32, _i, _vp]),
    "urhgpu_modulate_gfsk_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_uint32, _vp, _i, _f, _f, _f, _i, _vp, _i, _vp, _vp, _i64, C.POINTER(_i64)]),
    "urhgpu_modulate_gfsk": (_i, [_vp, _vp, _i64, C.c_uint32, _vp, _i, _f, _f, _f, C.c_uint32, C.c_uint32, _i, _vp, _i, _vp, _vp]),
    "urhgpu_spectrogram_dev": (_i, [_vp, _vp, _i64, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "urhgpu_bgra_lookup_dev": (_i, [_vp, _vp, _i64, _i, _vp, _i, _f, _f, _vp]),
    "urhgpu_convert_dev": (_i, [_vp, _vp, _i, _vp, _i, _i64]),
    "urhgpu_pcm_to_iq_dev": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "urhgpu_sub_encode_runs": (_i, [_vp, _i64, _i64, _vp, _i64, _vp]),
    "urhgpu_fft_peak_dev": (_i, [_vp, _vp, _i64, _vp]),
    "urhgpu_astype_dev": (_i, [_vp, _vp, _i, _vp, _i, _i64]),
    "urhgpu_path_minmax_dev": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _vp]),
    "urhgpu_path_minmax": (_i, [_vp, _vp, _i, _i64, _i64, _i64, _i64, _vp]),
    "urhgpu_bench_copy_ceiling_dev": (_i, [_vp, _vp, _vp, _i64, _i, _i, C.POINTER(_f)]),
    "urhgpu_memcpy_to_host": (_i, [_vp, _vp, _vp, _i64]),
    "urhgpu_memcpy_dtod": (_i, [_vp, _vp, _vp, _i64]),
    "urhgpu_test_force_state_bytes": (_i, [_i]),
    "urhgpu_test_force_tiles_per_chunk": (_i, [_i]),
    "urhgpu_test_wide_int_launches": (_i64, []),
    "urhgpu_test_force_generic_tail": (_i, [_i]),
    "urhgpu_test_atan2f_dev": (_i, [_vp, _vp, _vp, _i64, _vp]),
}

_lib = None
_lock = threading.Lock()


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) which libtorch_hip.so asks for by its UNVERSIONED file name, while liburhgpu.so
    asks for the SONAME: if liburhgpu.so is loaded before torch, the system runtime and torch's copy
    both end up in the process and the second one to initialise finds no device.  Pre-loading torch's
    copy (when torch is installed) makes liburhgpu.so's DT_NEEDED resolve to it by SONAME, whatever the
    import order; without torch the system runtime under /opt/rocm is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load liburhgpu.so; raises if it has not been built (python -m urh_amd.build)."""
    global _lib
    with _lock:
        if _lib is None:
            path = os.environ.get("URHGPU_LIB", LIB_PATH)      # A/B builds (python -m urh_amd.build --tag ...)
            if not os.path.exists(path):
                raise UrhGpuError(ERR_NO_DEVICE, f"{path} is missing: build it with `python -m urh_amd.build` "
                                                 "(there is no CPU fallback)")
            _share_torch_hip_runtime()
            lib = C.CDLL(path)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(lib, name)          # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def host_libm_check():
    """urhgpu_host_libm_check: {sincosf compared, mismatches, atan2f compared, mismatches} -- does the reference on THIS host call the libm
    the device code restates (include/urhgpu.h)?  Host arithmetic: works without a GPU."""
    out = (C.c_int64 * 4)()
    check(load().urhgpu_host_libm_check(out))
    return {"sincosf_compared": int(out[0]), "sincosf_mismatches": int(out[1]), "atan2f_compared": int(out[2]), "atan2f_mismatches": int(out[3])}


_libm_verdict = None


def host_libm_verdict(warn: bool = True):
    """host_libm_check(), once per process: a host whose libm is not the one the device code restates (a CPU without FMA: glibc then
    picks the other build of sinf / cosf) is told so -- the GPU results are then not bit-identical with what the reference computes ON
    THIS HOST for the Costas loop (about one sample in 10^9), though they still equal the reference run on an FMA host."""
    global _libm_verdict
    if _libm_verdict is None:
        _libm_verdict = host_libm_check()
        if warn and (_libm_verdict["sincosf_mismatches"] or _libm_verdict["atan2f_mismatches"]):
            import warnings
            warnings.warn("liburhgpu: this host's libm differs from the one the device code restates "
                          f"(sinf/cosf: {_libm_verdict['sincosf_mismatches']} of {_libm_verdict['sincosf_compared']}, atan2f: "
                          f"{_libm_verdict['atan2f_mismatches']} of {_libm_verdict['atan2f_compared']} probes): PSK / FSK results may differ from the "
                          "reference run on THIS host in isolated samples", RuntimeWarning, stacklevel=3)
    return _libm_verdict


def check(status: int):
    if status == OK:
        return
    lib = load()
    msg = lib.urhgpu_strerror(status).decode()
    if status == ERR_HIP:
        msg += ": " + lib.urhgpu_last_hip_error().decode()
    if status == ERR_DTYPE:
        raise ValueError("Unsupported dtype")      # same exception as the reference (signal_functions.pyx:283,354)
    raise UrhGpuError(status, msg)


class Context:
    """One GPU context (stream + scratch arena).  Not thread-safe: one per host thread."""

    def __init__(self, device: int = 0):
        lib = load()
        h = C.c_void_p()
        check(lib.urhgpu_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr):
        """Run on the given hipStream_t (0 / None = the HIP null stream, torch's default stream)."""
        check(load().urhgpu_ctx_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def use_private_stream(self):
        check(load().urhgpu_ctx_use_private_stream(self._h))

    def sync(self):
        check(load().urhgpu_ctx_sync(self._h))

    def set_pipelined(self, enable: bool, tail_stream_ptr=None):
        """see urhgpu_ctx_set_pipelined (include/urhgpu.h): outputs of iq_to_bits are then complete after join() / sync()"""
        check(load().urhgpu_ctx_set_pipelined(self._h, 1 if enable else 0, C.c_void_p(tail_stream_ptr or 0)))

    def set_tuning(self, key: str, value: int):
        """see urhgpu_ctx_set_tuning (include/urhgpu.h): tuning values of the pipelined mode (developer tooling)"""
        check(load().urhgpu_ctx_set_tuning(self._h, key.encode(), int(value)))

    def join(self):
        check(load().urhgpu_ctx_join(self._h))

    def reserve(self, n_samples: int, tolerance: int):
        check(load().urhgpu_ctx_reserve(self._h, int(n_samples), int(tolerance)))

    def info(self):
        cu, wf, mem = C.c_int(), C.c_int(), C.c_int64()
        name = C.create_string_buffer(256)
        check(load().urhgpu_ctx_info(self._h, C.byref(cu), C.byref(wf), C.byref(mem), name, 256))
        return {"compute_units": cu.value, "wavefront": wf.value, "hbm_bytes": mem.value, "name": name.value.decode()}

    def profile_begin(self, max_records: int):
        check(load().urhgpu_ctx_profile_begin(self._h, int(max_records)))

    def profile_end(self, cap: int = 4096):
        ms = (C.c_float * cap)()
        n = C.c_int(0)
        check(load().urhgpu_ctx_profile_end(self._h, ms, cap, C.byref(n)))
        return [float(ms[i]) for i in range(min(n.value, cap))]

    def costas_stats(self):
        """(chunks matched by a candidate, met at a checkpoint, evaluated serially, re-speculation rounds) of the last PSK pass"""
        out = (C.c_int32 * 4)()
        check(load().urhgpu_ctx_costas_stats(self._h, out))
        return tuple(int(v) for v in out)

    def close(self):
        if self._h:
            load().urhgpu_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context() -> Context:
    """Per-thread default context on device $URHGPU_DEVICE (default 0)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or ctx._h is None:
        ctx = Context(int(os.environ.get("URHGPU_DEVICE", "0")))
        _tls.ctx = ctx
    return ctx
