"""`Signal`-level shim: what the reference's `Signal` object promises about the demodulated signal, for a capture that lives in HBM.

Mirrors the behaviour (not the Qt plumbing) of /root/reference/src/urh/signalprocessing/Signal.py:

  * `qad` is computed lazily and cached (:421-431); captures that are already demodulated (mono WAV, Flipper `.sub`) bypass
    the demodulator -- their "qad" is the real part of the samples (:424-427);
  * `quad_demod()` returns `zeros(2)` when the noise threshold is at or above the largest possible magnitude of the sample
    type (:474-484), otherwise `afp_demod(iq, noise_threshold, modulation_type, 2 ** bits_per_symbol, costas_loop_bandwidth)`;
  * changing modulation_type / bits_per_symbol / costas_loop_bandwidth / noise_threshold drops the cache (:259, :273, :355, :391);
    center, center_spacing, tolerance, samples_per_symbol, pause_threshold do not (they only steer the slicing);
  * edits keep the cache consistent: insert -> dropped (:613-617), delete / crop -> sliced (:619-643), mute -> zeroed
    (:631-636), filter_range -> the range re-demodulated in place (:645-655).

Design difference: the capture and the cache are torch tensors on the GPU and one fused pass (urhgpu_iq_to_bits_dev) yields the
demodulated signal TOGETHER with the pulse table and the bits, so `get_protocol()` after a parameter change costs one pass
over the samples, not three.  Nothing here computes on the host: every entry point ends in liburhgpu.so and raises without it.
"""
import numpy as np

from . import _lib
from .pipeline import DemodParams, DevicePipeline, _torch_dtype

_DEMOD_KEYS = ("modulation_type", "bits_per_symbol", "costas_loop_bandwidth", "noise_threshold")
_SLICE_KEYS = ("center", "center_spacing", "tolerance", "samples_per_symbol", "pause_threshold", "message_length_divisor")


def _limits(np_dtype):
    """IQArray.min_max_for_dtype (IQArray.py:246-250)"""
    dt = np.dtype(np_dtype)
    if dt.kind in "fc":
        return -1, 1
    info = np.iinfo(dt)
    return info.min, info.max


class Signal:
    MODULATION_TYPES = ("ASK", "FSK", "PSK", "QAM")

    def __init__(self, iq=None, name="Signal", modulation="FSK", sample_rate=1e6, timestamp=0.0, pipe=None,
                 already_demodulated=False, device=None):
        """iq: (N, 2) numpy array / torch tensor of int8, uint8, int16, uint16 or float32, or complex64 (N,); uploaded once."""
        self.pipe = pipe or DevicePipeline(device)
        self.name = name
        self.sample_rate = float(sample_rate)
        self.timestamp = float(timestamp)
        self.already_demodulated = bool(already_demodulated)
        self.filename, self.wav_mode, self.flipper_raw_mode = "", False, False          # (Signal.py:69-70, :109)
        self._par = dict(modulation_type=modulation or "FSK", bits_per_symbol=1, costas_loop_bandwidth=0.1, noise_threshold=0,
                         center=0, center_spacing=1, tolerance=5, samples_per_symbol=100, pause_threshold=8,
                         message_length_divisor=1)
        self._qad = None
        self._bits = None            # BitsResult of the pass that produced _qad (valid for the slicing parameters in _bits_key)
        self._bits_key = None
        self.changed = False
        self.demod_passes = 0        # passes over the samples so far (tests watch the cache through it)
        self._iq = None
        if iq is not None:
            self.iq = iq

    # ---- samples --------------------------------------------------------------------------------------------------
    @classmethod
    def from_file(cls, filename, default_noise_threshold=None, **kw):
        """Signal(filename) (Signal.py:69-109): complex captures by extension (IQArray.py:205-227), `.coco` (a tar archive around one,
        :207-213), `.wav` (:114-173: 8 / 16 / 24 / 32-bit PCM; mono = already demodulated, stereo = I / Q; the file's sample rate becomes
        the Signal's) and Flipper `.sub` (:175-205: already demodulated).  default_noise_threshold: the reference's setting of that name
        (:97-107) -- "automatic": detect_noise_level of the magnitudes, a number: that many percent of max_magnitude; None (default here):
        the threshold stays 0 until the caller sets it."""
        from . import iq_array
        s = cls(None, **kw)
        s.wav_mode, s.flipper_raw_mode = filename.endswith(".wav"), filename.endswith(".sub")
        if s.wav_mode:
            iq, rate, demod = iq_array.from_wav(filename, device=s.pipe.device)
            s.already_demodulated = bool(demod)
            s.iq = iq
            s.sample_rate = float(rate)
        elif s.flipper_raw_mode:
            s.already_demodulated = True
            s.iq = iq_array.from_sub(filename, device=s.pipe.device)
        else:
            s.iq = iq_array.from_file(filename, device=s.pipe.device)
        s.filename = filename
        if default_noise_threshold == "automatic":
            from .estimators import detect_noise_level_dev
            s.noise_threshold = detect_noise_level_dev(s.pipe, s._iq)
        elif default_noise_threshold is not None:
            s.noise_threshold = float(default_noise_threshold) / 100 * s.max_magnitude
        return s

    def save(self):
        """Signal.save (:462-464)"""
        if self.changed:
            self.save_as(self.filename)

    def save_as(self, filename: str):
        """Signal.save_as (:466-472) -> FileOperator.save_signal (FileOperator.py:208-209): the capture converted on the device to what the
        extension names (raw sample types, `.wav`, `.coco`, `.sub`), written, the Signal renamed after the file"""
        import os
        from . import iq_array
        self.filename = filename
        iq_array.save_data(self._iq, filename, self.sample_rate)
        self.name = os.path.splitext(os.path.basename(filename))[0]
        self.changed = False

    @classmethod
    def from_file_streamed(cls, filename, pinned=None, **params):
        """`Signal(filename)` followed by `get_protocol_from_signal()` (Signal.py:42-112, IQArray.py:206-227, ProtocolAnalyzer.py:227-287)
        for a capture whose parameters are already known (a project file keeps them per signal): the file is read into PINNED host
        memory, and the upload does not wait for anything -- pieces are copied into the device buffer the Signal keeps and demodulated
        as they land (urhgpu_stream_push_upload); when the call returns the Signal holds the capture, its demodulated signal and the
        digitisation for the given parameters (`bits()` / `get_protocol()` cost nothing more).  params: the Signal's parameters
        (modulation_type, samples_per_symbol, center, tolerance, noise_threshold, ...) plus the constructor's keywords.
        Float32 / signed captures of ASK-less, PSK-less modulations take this route; everything else (unsigned sample types, which the
        reference converts first; ASK; PSK) falls back to from_file + the ordinary lazy passes -- same results either way.
        pinned: an optional dict that keeps the pinned read buffer AND the capture stream (three output slots, six pinned blobs: what a
        stream costs to build) between calls -- a file browser opening capture after capture with the same parameters; the stream is
        rebuilt when the parameters, the sample type or the size class change, `pinned["stream"].close()` releases it."""
        ctor = {k: params.pop(k) for k in ("name", "sample_rate", "timestamp", "pipe", "device") if k in params}
        if "modulation_type" in params:
            ctor["modulation"] = params.pop("modulation_type")
        signed = {".complex16s": np.int8, ".cs8": np.int8, ".complex32s": np.int16, ".cs16": np.int16}
        unsigned = (".complex16u", ".cu8", ".complex32u", ".cu16")
        if pinned is not None and ctor.get("pipe") is None:                      # the kept stream belongs to a pipeline: kept with it
            if pinned.get("pipe") is None:
                pinned["pipe"] = DevicePipeline(ctor.get("device"))
            ctor["pipe"] = pinned["pipe"]
        s = cls(None, **ctor)
        for k, v in params.items():
            setattr(s, k, v)
        dt = np.dtype(next((t for ext, t in signed.items() if filename.endswith(ext)), np.float32))
        lo, hi = _limits(dt)
        gated = not (s.noise_threshold < (2 * max(lo ** 2, hi ** 2)) ** 0.5)      # quad_demod's zeros(2) case (:474-484)
        if filename.endswith(unsigned) or s.modulation_type != "FSK" or gated:
            from .iq_array import from_file
            s.iq = from_file(filename, device=s.pipe.device)
            return s
        torch = s.pipe.torch
        import os
        n_values = os.path.getsize(filename) // dt.itemsize // 2 * 2          # convert_array_to_iq drops the last half sample (:238-239)
        n = n_values // 2
        if n < 3:
            from .iq_array import from_file
            s.iq = from_file(filename, device=s.pipe.device)
            return s
        keep = pinned if pinned is not None else {}
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16}[dt]
        buf = keep.get("buf")
        if buf is None or buf.dtype != tdt or buf.numel() < n_values:
            buf = torch.empty(n_values, dtype=tdt).pin_memory()
            keep["buf"] = buf
        host = buf[:n_values]
        with open(filename, "rb") as fh:                                       # straight into the pinned buffer: no pageable copy
            got = fh.readinto(memoryview(host.numpy()).cast("B"))
        if got != n_values * dt.itemsize:
            raise OSError(f"short read from {filename}")
        host = host.view(n, 2)
        dev = torch.empty((n, 2), dtype=tdt, device=s.pipe.device)
        # Whatever the streamed route cannot take -- a pulse table beyond the stream's default capacity (a noise-dominated capture, a low
        # noise threshold: ERR_CAPACITY / `truncated`), a capture of 2^31 samples and more or slicing parameters the stream rejects (ERR_ARG /
        # ERR_UNSUPPORTED and the reference's own ZeroDivisionError / OverflowError for them) -- falls back to the ordinary lazy passes with
        # their capacity retry, on the capture that is by then resident: same results either way.
        st = None
        pushed_before = 0
        kept = False                                                              # the stream lives on in `pinned`
        try:
            p = s.params()
            key = (id(s.pipe), repr(p), dt.str)
            if pinned is not None and keep.get("stream_key") == key and keep.get("stream") is not None and n <= keep.get("stream_n", 0):
                st, kept = keep["stream"], True                                   # (its results were copied out by the call that made them)
            else:
                if pinned is not None and keep.get("stream") is not None:
                    keep.pop("stream").close()
                    keep.pop("stream_key", None)
                st = s.pipe.stream(n, p, want_qad=True, want_pos=True, dtype=dt)      # (an upload: its own piece-wise route whatever the latency setting)
                if pinned is not None:
                    keep["stream"], keep["stream_key"], keep["stream_n"], kept = st, key, n, True
            pushed_before = st.stats()["pushed"]
            st.push_upload(host, dev)                                             # (host / dev stay referenced by this frame until flush() returns)
            (h,) = st.flush()
            h.check()
            from .pipeline import LazyDigitized
            qad = torch.empty(n, dtype=torch.float32, device=s.pipe.device)
            import ctypes as C
            _lib.check(_lib.load().urhgpu_memcpy_dtod(s.pipe.ctx.handle, C.c_void_p(qad.data_ptr()), C.c_void_p(h.d_qad_ptr), n * 4))
            s._iq = dev
            s._qad = qad
            s.demod_passes += 1
            s._bits, s._bits_key = LazyDigitized(h, want_pos=True).materialize(), s._slice_key()      # (the stream's pinned blob goes away with it)
        except (ZeroDivisionError, OverflowError, ValueError, _lib.UrhGpuError) as exc:
            if isinstance(exc, _lib.UrhGpuError) and exc.status not in (_lib.ERR_CAPACITY, _lib.ERR_ARG, _lib.ERR_UNSUPPORTED):
                raise
            s.pipe.ctx.sync()
            if st is None or st.stats()["pushed"] == pushed_before:
                dev.copy_(host, non_blocking=False)                               # the stream never took the capture: one plain copy
            s._iq = dev
            s._drop_cache()
            if kept and st is not None:                                           # a stream that failed is not kept
                keep.pop("stream", None); keep.pop("stream_key", None)
                kept = False
        finally:
            if st is not None and not kept:
                st.close()
        return s

    @property
    def iq(self):
        return self._iq

    @iq.setter
    def iq(self, value):
        torch = self.pipe.torch
        if isinstance(value, np.ndarray):
            if value.dtype == np.complex64:
                value = value.view(np.float32).reshape(-1, 2)
            value = torch.from_numpy(np.ascontiguousarray(value))
        if value.dtype == torch.complex64:
            value = torch.view_as_real(value)
        if value.dim() == 1:                      # IQArray.convert_array_to_iq: flat interleaved, odd tail dropped (:229-243)
            value = value[: value.numel() // 2 * 2].reshape(-1, 2)
        _torch_dtype(value)                        # ValueError("Unsupported dtype") as the Cython signatures give
        self._iq = value.to(self.pipe.device).contiguous()
        self._drop_cache()

    @property
    def num_samples(self):
        return 0 if self._iq is None else int(self._iq.shape[0])

    @property
    def dtype(self):
        return _torch_dtype(self._iq)

    @property
    def max_magnitude(self):                       # Signal.py:407-410
        lo, hi = _limits(self.dtype)
        return (2 * max(lo ** 2, hi ** 2)) ** 0.5

    @property
    def max_amplitude(self):                       # :412-415
        lo, hi = _limits(self.dtype)
        return 0.5 * (hi - lo)

    @property
    def noise_threshold_relative(self):
        return self.noise_threshold / self.max_magnitude

    @noise_threshold_relative.setter
    def noise_threshold_relative(self, value):
        self.noise_threshold = value * self.max_magnitude

    @property
    def modulation_order(self):
        return 2 ** self.bits_per_symbol

    # ---- parameters: one generic accessor pair, the key decides what a change invalidates ----------------------------
    def __getattr__(self, key):
        par = self.__dict__.get("_par")
        if par is not None and key in par:
            return par[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        par = self.__dict__.get("_par")
        if par is not None and key in par:
            if key == "bits_per_symbol":
                value = int(value)
            if par[key] != value:
                par[key] = value
                if key in _DEMOD_KEYS:
                    self._drop_cache()
            return
        object.__setattr__(self, key, value)

    def params(self) -> DemodParams:
        p = self._par
        return DemodParams(p["modulation_type"], int(p["bits_per_symbol"]), float(p["noise_threshold"]), float(p["center"]),
                           float(p["center_spacing"]), int(p["tolerance"]), int(p["samples_per_symbol"]),
                           float(p["costas_loop_bandwidth"]), int(p["pause_threshold"]), True)

    def _drop_cache(self, release: bool = False):
        self._qad = None
        self._bits = None
        self._bits_key = None
        # the digitisations' views of the pinned host buffers go with the cache (a digitisation somebody still holds is widened first); the
        # buffers themselves stay -- a noise threshold dragged through its range in the GUI lands here on every change, and pinning memory
        # again each time is a hipHostMalloc per change -- and are released with the samples (eliminate)
        users = self.__dict__.get("_host_users")
        if users is not None:
            for k, ref in enumerate(users):
                old = ref() if ref is not None else None
                if old is not None:
                    old.materialize()
                users[k] = None
            for pool in self.__dict__.get("_host_pools", ()):
                if release:
                    pool.clear()
                else:
                    pool["owner"] = None

    def _slice_key(self):
        return tuple(self._par[k] for k in _SLICE_KEYS[:5])

    # ---- demodulated signal -------------------------------------------------------------------------------------------
    @property
    def real_plot_data(self):
        return self._iq[:, 0] if self._iq is not None else self.pipe.torch.zeros(0, dtype=self.pipe.torch.float32, device=self.pipe.device)

    @property
    def imag_plot_data(self):
        return self._iq[:, 1] if self._iq is not None else self.pipe.torch.zeros(0, dtype=self.pipe.torch.float32, device=self.pipe.device)

    # ---- what the reference's dialogs ask a Signal besides its bits ----------------------------------------------------------
    def get_thresholds_for_center(self, center: float, spacing=None):
        """Signal.get_thresholds_for_center (:531-535)"""
        from .signal_functions import get_center_thresholds
        return get_center_thresholds(center, self.center_spacing if spacing is None else spacing, self.modulation_order)

    @property
    def center_thresholds(self):
        return self.get_thresholds_for_center(self.center)

    def calc_relative_noise_threshold_from_range(self, noise_start: int, noise_end: int):
        """Signal.calc_relative_noise_threshold_from_range (:486-506): the largest normalised magnitude of the selected range, rounded up to
        four digits -- one pass of the magnitude statistics kernel over the range (urhgpu_magnitude_chunk_stats_dev), the arithmetic on its
        ONE number as numpy does it (float32 magnitude / float64 norm).  An empty range: the current relative threshold, as the reference."""
        import ctypes as C
        num_digits = 4
        noise_start, noise_end = int(noise_start), int(noise_end)
        if noise_start > noise_end:
            noise_start, noise_end = noise_end, noise_start
        n = self.num_samples
        lo = max(noise_start + n, 0) if noise_start < 0 else min(noise_start, n)       # (numpy slice semantics of subarray(start, stop))
        hi = max(noise_end + n, 0) if noise_end < 0 else min(noise_end, n)
        if hi <= lo:
            return self.noise_threshold_relative
        torch = self.pipe.torch
        rng = self._iq[lo:hi]
        out = torch.empty(2, dtype=torch.float64, device=self.pipe.device)
        from .iq_array import _DT
        self.pipe.ctx.set_stream(torch.cuda.current_stream(self.pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_magnitude_chunk_stats_dev(self.pipe.ctx.handle, C.c_void_p(rng.data_ptr()), _DT[_torch_dtype(rng)], hi - lo, hi - lo, 1,
                                                                C.c_void_p(out[0:1].data_ptr()), C.c_void_p(out[1:2].data_ptr())))
        mi, ma = _limits(self.dtype)
        maximum = np.float64(np.float32(out[1].item())) / np.sqrt(ma ** 2.0 + mi ** 2.0)
        return np.ceil(maximum * 10 ** num_digits) / 10 ** num_digits

    def estimate_frequency(self, start: int, end: int, sample_rate: float):
        """Signal.estimate_frequency (:578-601): the frequency of the strongest FFT bin over the largest power-of-two window that fits
        [start, end), in Hertz (absolute value) -- urhgpu_fft_peak_dev on the window's complex64 conversion; the reference's fallback of
        100 kHz where its FFT raises (an empty or reversed range)."""
        import ctypes as C
        import math
        try:
            length = 2 ** int(math.log2(end - start))
        except ValueError:
            return 100e3
        from .iq_array import convert_to
        data = convert_to(self._iq[start:start + length], np.float32, self.pipe.ctx).contiguous()
        n = int(data.shape[0])
        if n == 0:
            return 100e3                                      # (np.fft.fft of an empty array: ValueError in the reference)
        n_fft = n if (n & (n - 1)) == 0 else None
        if n_fft is None:                                     # the window ran past the end of the capture: numpy transforms what is there;
            raise ValueError("estimate_frequency: the window [start, start + 2^k) must lie inside the capture")   # not a power of two -- not built
        peak = C.c_int64(0)
        torch = self.pipe.torch
        self.pipe.ctx.set_stream(torch.cuda.current_stream(self.pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_fft_peak_dev(self.pipe.ctx.handle, C.c_void_p(data.data_ptr()), n, C.byref(peak)))
        idx = int(peak.value)
        freq = idx / n if idx < (n + 1) // 2 else (idx - n) / n          # np.fft.fftfreq(n)[idx]
        return abs(freq * sample_rate)

    def create_new(self, start=0, end=0, new_data=None, new_timestamp=0):
        """Signal.create_new (:508-529): a Signal over a slice of this capture (a device copy) or over new samples, carrying this one's
        noise threshold, samples_per_symbol, bits_per_symbol, center, sample rate and file-mode flags; marked changed."""
        new = Signal(None, name="New " + self.name, sample_rate=self.sample_rate, pipe=self.pipe, already_demodulated=self.already_demodulated)
        if new_data is None:
            new.iq = self._iq[start:end].clone()
            new.timestamp = self.timestamp + (start / self.sample_rate)
        else:
            new.iq = new_data
            new.timestamp = new_timestamp
        for k in ("noise_threshold", "samples_per_symbol", "bits_per_symbol", "center"):
            setattr(new, k, self._par[k])
        new.wav_mode, new.flipper_raw_mode = self.wav_mode, self.flipper_raw_mode
        new.changed = True
        return new

    @staticmethod
    def from_samples(samples, name: str, sample_rate: float, pipe=None):
        """Signal.from_samples (:659-664)"""
        return Signal(samples, name=name, sample_rate=sample_rate, pipe=pipe)

    def silent_set_modulation_type(self, mod_type: str):
        """Signal.silent_set_modulation_type (:608-609): no invalidation (the caller knows the cache is still right)"""
        self._par["modulation_type"] = mod_type

    def eliminate(self):
        """Signal.eliminate (:603-606): drop the samples and everything derived from them"""
        self._iq = None
        self._drop_cache(release=True)

    def quad_demod(self):
        """Signal.quad_demod (:474-484): a fresh demodulation (device tensor), or zeros(2) when everything is below the noise gate."""
        torch = self.pipe.torch
        if not (self.noise_threshold < self.max_magnitude):
            return torch.zeros(2, dtype=torch.float32, device=self.pipe.device)
        self.demod_passes += 1
        return self.pipe.afp_demod(self._iq, self.params())

    @property
    def qad(self):
        """Signal.qad (:421-431): cached; already-demodulated captures hand back their real part."""
        if self._qad is None:
            torch = self.pipe.torch
            if self.already_demodulated:
                self._qad = self.real_plot_data.contiguous()
            elif not (self.noise_threshold < self.max_magnitude) or self.num_samples <= 2 or self.modulation_type not in ("ASK", "FSK"):
                self._qad = self.quad_demod()
            else:
                # ASK / FSK: the fused pass gives qad AND the pulse table / bits for the current slicing parameters
                # (the reference's qad depends on the demodulation parameters only: slicing parameters the fused pass cannot take
                # -- samples_per_symbol < 1, tolerance outside uint16, a pulse table beyond the output capacity -- must not make
                # the qad fail; they raise where the reference raises, in ppseq() / bits())
                res = None
                try:
                    res = self.pipe.iq_to_bits_checked(self._iq, self.params(), want_qad=True)
                except (ZeroDivisionError, OverflowError, ValueError):
                    pass
                except _lib.UrhGpuError as e:
                    if e.status not in (_lib.ERR_CAPACITY, _lib.ERR_ARG, _lib.ERR_UNSUPPORTED):
                        raise
                if res is None:
                    self._qad = self.quad_demod()
                else:
                    self.demod_passes += 1
                    self._qad = res.qad.clone()          # the pipeline's buffers are reused by the next pass
                    self._bits, self._bits_key = self._detach(res), self._slice_key()
        return self._qad

    def _detach(self, res):
        """The pass's results on the host (the pipeline's output buffers are overwritten by the next pass): ONE pinned copy of the compact
        blob (BitsResult.host) into a buffer of this Signal's; the reference-shaped arrays -- int64 pulse table, a byte per bit, int64
        positions -- are made from it when somebody asks (LazyDigitized).  Two pinned buffers alternate: a digitisation that is still
        referenced when its buffer comes up for reuse is widened first."""
        import weakref
        from .pipeline import LazyDigitized
        pools = self.__dict__.setdefault("_host_pools", [{}, {}])
        users = self.__dict__.setdefault("_host_users", [None, None])
        k = self.__dict__.get("_host_turn", 0)
        self.__dict__["_host_turn"] = k ^ 1
        old = users[k]() if users[k] is not None else None
        if old is not None:
            old.materialize()
        h = res.host(pool=pools[k])
        if h.truncated & 4:                       # a value the blob's narrow types cannot hold (a capture shorter than the tolerance): wide copies
            return (res.ppseq().copy(),) + tuple(x.copy() for x in res.flat())
        out = LazyDigitized(h, want_pos=res.pos_buf is not None)
        users[k] = weakref.ref(out)
        return out

    def qad_host(self) -> np.ndarray:
        return self.qad.cpu().numpy()

    # ---- digitisation ---------------------------------------------------------------------------------------------------
    def _digitize(self):
        """(ppseq, bits, msg_off, pauses, pos, pos_off) for the current parameters; re-slices the cached qad when only slicing
        parameters changed (grab_pulse_lens + _ppseq_to_bits on the device, no new demodulation)."""
        import ctypes as C
        q = self.qad
        if self._bits is not None and self._bits_key == self._slice_key():
            return self._bits
        torch = self.pipe.torch
        if q.dtype != torch.float32:
            raise ValueError("Buffer dtype mismatch, expected 'float' (grab_pulse_lens takes float[::1])")
        p = self.params()
        n = int(q.shape[0])
        if n == 0:
            z = np.zeros(0, np.int64)
            return (np.zeros((0, 2), np.int64), np.zeros(0, np.uint8), np.zeros(1, np.int64), z, z, np.zeros(1, np.int64))
        res = self._digitize_dev(q, p)
        out = self._detach(res)
        self._bits, self._bits_key = out, self._slice_key()
        return out

    def _digitize_dev(self, q, p):
        """The device part of _digitize: grab_pulse_lens + _ppseq_to_bits on the cached qad, outputs left in the pipeline's buffers
        (no host synchronisation): a BitsResult."""
        return self.pipe.qad_to_bits(q, p, slot=7)

    def ppseq(self) -> np.ndarray:
        """signal_functions.grab_pulse_lens(signal.qad, center, tolerance, modulation_type, samples_per_symbol, bits_per_symbol,
        center_spacing) as ProtocolAnalyzer.get_protocol_from_signal calls it (ProtocolAnalyzer.py:236-244)."""
        return self._digitize()[0]

    def bits(self):
        """(bit_data, pauses, bit_sample_pos) exactly as ProtocolAnalyzer._ppseq_to_bits returns them (:323-414)."""
        import array
        _, b, off, pauses, pos, poff = self._digitize()
        data = [array.array("B", b[off[i]:off[i + 1]].tobytes()) for i in range(len(pauses))]
        return data, array.array("L", pauses.tolist()), [array.array("L", pos[poff[i]:poff[i + 1]].tolist()) for i in range(len(pauses))]

    def plain_bits_str(self):
        return ["".join(map(str, m)) for m in self.bits()[0]]

    def get_protocol(self):
        """ProtocolAnalyzer.get_protocol_from_signal (:227-287): MessageData per message (bits, pause, RSSI, timestamp, positions)."""
        from .protocol import messages_from_bits
        data, pauses, bsp = self.bits()
        return messages_from_bits(self.pipe, self._iq, self.params(), data, pauses, bsp, int(self.message_length_divisor),
                                  self.sample_rate, self.timestamp)

    # ---- parameter estimation -----------------------------------------------------------------------------------------------
    def auto_detect(self, detect_modulation=True, detect_noise=False) -> bool:
        """Signal.auto_detect (:537-577): AutoInterpretation.estimate on the device-resident capture, parameters applied as the
        reference applies them (noise and modulation only when asked for; center, tolerance, samples_per_symbol always)."""
        from .estimators import estimate_dev
        modulation = None if detect_modulation else ("OOK" if self.bits_per_symbol == 1 and self.modulation_type == "ASK"
                                                     else self.modulation_type)
        keep = {}
        est = estimate_dev(self.pipe, self._iq, noise=None if detect_noise else self.noise_threshold, modulation=modulation, keep=keep)
        if est is None:
            return False
        if detect_noise:
            self.noise_threshold = est["noise"]
        if detect_modulation:
            self.modulation_type = est["modulation_type"]
        self.center = est["center"]
        self.tolerance = est["tolerance"]
        self.samples_per_symbol = est["bit_length"]
        # estimate demodulated the capture with afp_demod(iq, noise, mod, 2) (AutoInterpretation.py:397-402): when that is what
        # Signal.qad would compute for the parameters just applied, it IS the cache (the reference demodulates once more here)
        q = keep.get("qad")
        if q is not None and self._qad is None and not self.already_demodulated and int(q.shape[0]) == self.num_samples and \
                keep["mod"] == self.modulation_type and self.bits_per_symbol == 1 and \
                np.float32(keep["noise"]) == np.float32(self.noise_threshold) and self.noise_threshold < self.max_magnitude and \
                (keep["mod"] != "PSK" or np.float32(self.costas_loop_bandwidth) == np.float32(0.1)) and self.num_samples > 2:
            self._qad = q
        return True

    # ---- edits ------------------------------------------------------------------------------------------------------------
    def _after_edit(self):
        self._bits = None
        self._bits_key = None
        self.changed = True

    def insert_data(self, index: int, data):
        torch = self.pipe.torch
        d = torch.from_numpy(np.ascontiguousarray(data)).to(self.pipe.device) if isinstance(data, np.ndarray) else data
        self._iq = torch.cat([self._iq[:index], d.to(self._iq.dtype).reshape(-1, 2), self._iq[index:]]).contiguous()
        self._qad = None
        self._after_edit()

    def delete_range(self, start: int, end: int):
        torch = self.pipe.torch
        n_before = self.num_samples
        self._iq = torch.cat([self._iq[:start], self._iq[end:]]).contiguous()
        # the reference indexes the cache with a mask of num_samples entries: a zeros(2) cache (noise threshold at the sample
        # type's maximum) does not take it -- IndexError, logged, cache kept as it is (Signal.py:619-629)
        if self._qad is not None and int(self._qad.shape[0]) == n_before:
            self._qad = torch.cat([self._qad[:start], self._qad[end:]]).contiguous()
        self._after_edit()

    def mute_range(self, start: int, end: int):
        self._iq[start:end] = 0
        if self._qad is not None:
            self._qad[start:end] = 0
        self._after_edit()

    def crop_to_range(self, start: int, end: int):
        self._iq = self._iq[start:end].contiguous()
        if self._qad is not None:
            self._qad = self._qad[start:end].contiguous()
        self._after_edit()

    def filter_range(self, start: int, end: int, taps):
        """Signal.filter_range (:645-655) with Filter.work's FIR branch (Filter.py:31-46): the range is filtered on its own (zero
        history), written back in the capture's sample type, and qad[start:end] becomes afp_demod of the filtered range ALONE
        (so its first sample is the NOISE value, as in the reference)."""
        import ctypes as C
        torch = self.pipe.torch
        _ = self.qad                                  # the reference indexes self._qad: it must exist
        seg = self._iq[start:end]
        n = int(seg.shape[0])
        if n == 0:
            return
        # the RAW sample values as float32 (Filter.apply_fir_filter, Filter.py:37-41: no IQArray scaling), filtered, and written
        # back with numpy's truncating cast (IQArray.__setitem__, IQArray.py:31-33)
        from .iq_array import astype
        x = astype(seg.clone(), np.float32, self.pipe.ctx)              # a copy: 16-byte aligned whatever `start` is
        h = np.ascontiguousarray(taps, dtype=np.complex64)
        d_h = torch.from_numpy(h.view(np.float32).copy()).to(self.pipe.device)
        y = torch.empty_like(x)
        self.pipe.ctx.set_stream(torch.cuda.current_stream(self.pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_fir_filter_dev(self.pipe.ctx.handle, C.c_void_p(x.data_ptr()), n, C.c_void_p(d_h.data_ptr()),
                                                     len(h), None, C.c_void_p(y.data_ptr())))
        if seg.dtype != torch.float32:
            y = astype(y, self.dtype, self.pipe.ctx)
        self._iq[start:end] = y
        if self._qad.shape[0] == self.num_samples:        # (a zeros(2) cache cannot take the range: the reference raises there too)
            self.demod_passes += 1
            self._qad[start:end] = self.pipe.afp_demod(self._iq[start:end].clone(), self.params())
        else:
            raise ValueError("could not broadcast the demodulated range into a cache of shape ({},)".format(int(self._qad.shape[0])))
        self._after_edit()
