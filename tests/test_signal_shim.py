"""`urh_amd.signal.Signal` against the reference's Signal semantics (Signal.py:421-431, 474-484, 259/273/355/391,
613-655): lazy qad cache, invalidation, zeros(2) rule, already-demodulated bypass, edits, filter_range -- checked against the
oracle and, where oracle/_ref holds the staged reference, against the real Signal / ProtocolAnalyzer objects step by step."""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden, synth_fsk

pytestmark = pytest.mark.gpu


def u32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def pipe():
    from urh_amd.pipeline import DevicePipeline
    return DevicePipeline()


def _apply(sig, g):
    sig.modulation_type = g["modulation_type"]
    sig.bits_per_symbol = g["bits_per_symbol"]
    sig.noise_threshold = g["noise_threshold"]
    sig.center = g["center"]
    sig.center_spacing = g["center_spacing"]
    sig.tolerance = g["tolerance"]
    sig.samples_per_symbol = g["samples_per_symbol"]
    sig.pause_threshold = g["pause_threshold"]
    sig.costas_loop_bandwidth = g["costas_loop_bandwidth"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_signal_goldens(pipe, name):
    """Every golden capture through the Signal object: qad, pulse table and bit strings as the reference recorded them."""
    from urh_amd.signal import Signal
    g = load_golden(name)
    sig = Signal(g["iq"], pipe=pipe)
    _apply(sig, g)
    first = 1 if g["modulation_type"] == "PSK" else 0          # reference leaves qad[0] uninitialised for PSK
    assert np.array_equal(u32(sig.qad_host())[first:], u32(g["qad"])[first:])
    assert sig.demod_passes == 1
    if first:
        return          # the golden table was sliced from a qad whose element 0 was whatever np.empty held in the reference run
    assert np.array_equal(sig.ppseq(), g["ppseq"])
    bits, off = g["bits"], g["msg_off"]
    want = ["".join(map(str, bits[off[i]:off[i + 1]])) for i in range(len(off) - 1)]
    assert sig.plain_bits_str() == want
    assert sig.demod_passes == 1


def test_cache_and_invalidation(pipe, oracle):
    from urh_amd.signal import Signal
    iq = synth_fsk(300_000, sps=100, seed=3, noise=0.05, pause_every=90_000, pause_len=6000)
    sig = Signal(iq, pipe=pipe)
    sig.center = 0.0
    q0 = sig.qad_host()
    assert sig.demod_passes == 1
    assert np.array_equal(u32(q0), u32(oracle.afp_demod(iq, 0.0, "FSK", 2)))
    b0 = sig.plain_bits_str()
    _ = sig.qad
    assert sig.demod_passes == 1                                   # cached (Signal.py:421-431); bits came with the same pass
    # slicing parameters: no new demodulation (Signal.py:297-340 do not touch _qad)
    for key, val in (("center", 0.05), ("tolerance", 2), ("samples_per_symbol", 50), ("pause_threshold", 4), ("center_spacing", 0.5)):
        setattr(sig, key, val)
        pp = oracle.grab_pulse_lens(q0, sig.center, sig.tolerance, "FSK", sig.samples_per_symbol, 1, sig.center_spacing)
        assert np.array_equal(sig.ppseq(), pp), key
        fb = oracle.ppseq_to_bits_flat(pp, sig.samples_per_symbol, 1, True, sig.pause_threshold)
        data, pauses, bsp = sig.bits()
        assert np.array_equal(np.concatenate([np.frombuffer(d, np.uint8) for d in data]) if data else np.zeros(0, np.uint8), fb[0]), key
        assert list(pauses) == fb[2].tolist(), key
        assert sig.demod_passes == 1, key
    # demodulation parameters: cache dropped (:259, :273, :355, :391); same value again: kept
    passes = 1
    for key, val in (("noise_threshold", 0.2), ("modulation_type", "ASK"), ("bits_per_symbol", 2), ("costas_loop_bandwidth", 0.05)):
        setattr(sig, key, val)
        assert sig._qad is None, key
        q = sig.qad_host()
        passes += 1
        assert sig.demod_passes == passes, key
        setattr(sig, key, val)
        assert sig._qad is not None and sig.demod_passes == passes, key
        assert np.array_equal(u32(q), u32(oracle.afp_demod(iq, sig.noise_threshold, sig.modulation_type, sig.modulation_order))), key
    assert b0 != [] and sig.plain_bits_str() is not None


@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.uint8, np.int16, np.uint16])
def test_zeros2_rule(pipe, dtype):
    """quad_demod: noise_threshold >= max_magnitude -> np.zeros(2) (Signal.py:475-484), for every sample type's max_magnitude."""
    from urh_amd.signal import Signal
    iq = synth_fsk(5000, seed=1, dtype=dtype)
    sig = Signal(iq, pipe=pipe)
    want_max = {np.float32: 2 ** 0.5, np.int8: (2 * 128 ** 2) ** 0.5, np.uint8: (2 * 255 ** 2) ** 0.5,
                np.int16: (2 * 32768 ** 2) ** 0.5, np.uint16: (2 * 65535 ** 2) ** 0.5}[dtype]
    assert sig.max_magnitude == want_max
    sig.noise_threshold = sig.max_magnitude
    q = sig.qad_host()
    assert q.dtype == np.float32 and q.shape == (2,) and not q.any()
    assert sig.demod_passes == 0
    sig.noise_threshold = np.nextafter(np.float64(sig.max_magnitude), 0)
    assert sig.qad_host().shape == (5000,) and sig.demod_passes == 1
    sig.noise_threshold_relative = 1.5
    assert sig.qad_host().shape == (2,)


def test_already_demodulated_bypass(pipe, oracle):
    """mono WAV / Flipper .sub captures (Signal.py:424-427): qad is the real part, the demodulator never runs."""
    from urh_amd.signal import Signal
    rng = np.random.default_rng(2)
    rect = np.repeat(rng.integers(0, 2, 400), 50).astype(np.float32) * 0.8 + 0.1 + 0.01 * rng.standard_normal(20_000).astype(np.float32)
    iq = np.stack([rect, np.zeros_like(rect)], 1)
    sig = Signal(iq, pipe=pipe, already_demodulated=True, modulation="ASK")
    sig.center, sig.samples_per_symbol = 0.5, 50
    assert np.array_equal(u32(sig.qad_host()), u32(rect))
    pp = oracle.grab_pulse_lens(rect, 0.5, 5, "ASK", 50, 1, 1.0)
    assert np.array_equal(sig.ppseq(), pp)
    sig.noise_threshold = 0.3                   # would change a real demodulation; here the real part is handed back again
    assert np.array_equal(u32(sig.qad_host()), u32(rect))
    assert sig.demod_passes == 0


def test_edits_and_filter_range(pipe, oracle):
    from urh_amd.signal import Signal
    iq = synth_fsk(120_000, sps=100, seed=8, noise=0.03)
    sig = Signal(iq, pipe=pipe)
    sig.noise_threshold = 0.1
    q = sig.qad_host().copy()
    host = iq.copy()
    sig.mute_range(1000, 3000)                                  # :631-636 both zeroed, no new pass
    host[1000:3000] = 0
    q[1000:3000] = 0
    assert np.array_equal(u32(sig.qad_host()), u32(q)) and sig.demod_passes == 1
    sig.delete_range(50_000, 60_001)                            # :619-629 both sliced
    host = np.concatenate([host[:50_000], host[60_001:]])
    q = np.concatenate([q[:50_000], q[60_001:]])
    assert sig.num_samples == len(host) and np.array_equal(u32(sig.qad_host()), u32(q)) and sig.demod_passes == 1
    assert np.array_equal(sig.ppseq(), oracle.grab_pulse_lens(q, 0.0, 5, "FSK", 100, 1, 1.0))
    # filter_range (:645-655): FIR on the range alone, then afp_demod of the range alone written into the cache
    taps = (np.hanning(33) / np.hanning(33).sum()).astype(np.complex64)
    a, b = 20_003, 47_777
    sig.filter_range(a, b, taps)
    seg = np.ascontiguousarray(host[a:b]).view(np.complex64).reshape(-1)
    filt = oracle.fir_filter(seg, taps).view(np.float32).reshape(-1, 2)
    host[a:b] = filt
    q[a:b] = oracle.afp_demod(np.ascontiguousarray(host[a:b]), np.float32(0.1), "FSK", 2)
    assert np.array_equal(sig.iq.cpu().numpy().view(np.uint32), host.view(np.uint32))
    assert np.array_equal(u32(sig.qad_host()), u32(q))
    sig.crop_to_range(10_000, 90_000)                           # :638-643
    assert np.array_equal(u32(sig.qad_host()), u32(q[10_000:90_000])) and sig.num_samples == 80_000
    sig.insert_data(5, np.zeros((7, 2), np.float32))            # :613-617 cache dropped
    assert sig._qad is None and sig.num_samples == 80_007


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16])
def test_filter_range_on_integer_captures(pipe, oracle, dtype):
    """Signal.filter_range on integer captures: the reference filters the RAW integer values as complex64 (Filter.py:37-41) and
    writes them back with numpy's truncating cast (IQArray.__setitem__, IQArray.py:31-33) -- no IQArray scaling either way.
    Against the real Signal / Filter objects where oracle/_ref holds them, else against the same steps in numpy."""
    import ref_python
    from urh_amd.signal import Signal
    iq = synth_fsk(60_000, sps=100, seed=21, noise=0.03, dtype=dtype)
    taps = (np.hanning(17) / np.hanning(17).sum() * 0.9).astype(np.complex64)
    a, b = 10_001, 41_234
    mine = Signal(iq, pipe=pipe)
    mine.noise_threshold = 3.0
    _ = mine.qad
    mine.filter_range(a, b, taps)
    if ref_python.available():
        ref_python.setup()
        from urh.signalprocessing.Filter import Filter
        from urh.signalprocessing.IQArray import IQArray
        from urh.signalprocessing.Signal import Signal as RefSignal
        ref = RefSignal("")
        ref.iq_array = IQArray(iq.copy())
        ref.noise_threshold = 3.0
        _ = ref.qad
        ref.filter_range(a, b, Filter(list(taps)))
        want_iq, want_qad = ref.iq_array.data, np.asarray(ref.qad)
    else:
        seg = np.empty(b - a, np.complex64)
        seg.real, seg.imag = iq[a:b, 0], iq[a:b, 1]
        f = oracle.fir_filter(seg, taps)
        want_iq = iq.copy()
        want_iq[a:b, 0], want_iq[a:b, 1] = f.real, f.imag
        want_qad = oracle.afp_demod(iq, np.float32(3.0), "FSK", 2)
        want_qad[a:b] = oracle.afp_demod(np.ascontiguousarray(want_iq[a:b]), np.float32(3.0), "FSK", 2)
    assert np.array_equal(mine.iq.cpu().numpy(), want_iq)
    assert np.array_equal(u32(mine.qad_host()), u32(want_qad))


def test_qad_survives_bad_slicing_parameters(pipe, oracle):
    """The reference's qad depends on the demodulation parameters only (Signal.py:421-431): samples_per_symbol = 0 or a tolerance
    outside uint16 make grab_pulse_lens / _ppseq_to_bits fail, not the qad; delete_range leaves a zeros(2) cache alone (:619-629)."""
    from urh_amd.signal import Signal
    iq = synth_fsk(20_000, sps=100, seed=5, noise=0.02)
    sig = Signal(iq, pipe=pipe)
    sig.samples_per_symbol = 0
    assert np.array_equal(u32(sig.qad_host()), u32(oracle.afp_demod(iq, 0.0, "FSK", 2)))
    with pytest.raises(ZeroDivisionError):
        sig.bits()
    sig2 = Signal(iq, pipe=pipe)
    sig2.tolerance = 70_000
    assert np.array_equal(u32(sig2.qad_host()), u32(oracle.afp_demod(iq, 0.0, "FSK", 2)))
    with pytest.raises(OverflowError):
        sig2.ppseq()
    sig3 = Signal(iq, pipe=pipe)
    sig3.noise_threshold = sig3.max_magnitude
    assert sig3.qad_host().shape == (2,)
    sig3.delete_range(100, 200)
    assert sig3.num_samples == 19_900 and sig3.qad_host().shape == (2,)


def test_against_the_real_reference_signal_object(pipe):
    """The same sequence of parameter changes on the real reference's Signal + ProtocolAnalyzer (staged under oracle/_ref) and on
    urh_amd.signal.Signal: qad and bit strings equal after every step."""
    import ref_python
    if not ref_python.available():
        pytest.skip("oracle/_ref (compiled reference + Python sources) not present")
    ref_python.setup()
    from urh.signalprocessing.IQArray import IQArray
    from urh.signalprocessing.ProtocolAnalyzer import ProtocolAnalyzer
    from urh.signalprocessing.Signal import Signal as RefSignal
    from urh_amd.signal import Signal
    for name in ("fsk", "ask", "homematic_i16", "two_participants_i8"):
        g = load_golden(name)
        ref = RefSignal("")
        ref.iq_array = IQArray(g["iq"])
        mine = Signal(g["iq"], pipe=pipe)
        steps = [("modulation_type", g["modulation_type"]), ("samples_per_symbol", g["samples_per_symbol"]), ("center", g["center"]),
                 ("noise_threshold", g["noise_threshold"]), ("tolerance", g["tolerance"]), ("tolerance", 1),
                 ("noise_threshold", float(g["noise_threshold"]) * 1.5 + 0.01), ("center", g["center"] * 0.9),
                 ("noise_threshold", mine.max_magnitude), ("noise_threshold", g["noise_threshold"]),
                 ("modulation_type", "ASK" if g["modulation_type"] == "FSK" else "FSK"), ("bits_per_symbol", 2), ("center_spacing", 0.3)]
        for key, val in steps:
            setattr(ref, key, val)
            setattr(mine, key, val)
            rq = np.asarray(ref.qad)
            assert np.array_equal(u32(mine.qad_host()), u32(rq)), (name, key, val)
            if len(rq) > 2:
                pa = ProtocolAnalyzer(ref)
                pa.get_protocol_from_signal()
                assert mine.plain_bits_str() == pa.plain_bits_str, (name, key, val)
                msgs = mine.get_protocol()
                assert [m.pause for m in msgs] == [m.pause for m in pa.messages], (name, key, val)
                assert [list(m.bit_sample_pos) for m in msgs] == [list(m.bit_sample_pos) for m in pa.messages], (name, key, val)


@pytest.mark.parametrize("ext,dtype,n", [(".complex", np.float32, 1 << 23), (".complex", np.float32, (1 << 20) - 777), (".complex32s", np.int16, 1 << 23),
                                         (".cs8", np.int8, 3 << 21), (".cu8", np.uint8, 1 << 19)])
def test_from_file_streamed_equals_from_file(oracle, tmp_path, ext, dtype, n):
    """Signal.from_file_streamed -- the file read into pinned memory, uploaded piece by piece and demodulated as it lands -- leaves the
    Signal in the state Signal.from_file + qad + bits() reach lazily: the capture on the device, the demodulated signal, the pulse table,
    bits, pauses, positions; all of it equal to the oracle.  (2^23 samples of whole tiles: the segmented upload, four pieces; a partial
    tile or fewer than 512 chunks: one copy + an ordinary pass; unsigned samples: the from_file fallback, converted on the device as the reference's IQArray does on the host.)"""
    from urh_amd.signal import Signal
    iq = synth_fsk(n, sps=100, seed=31, noise=0.04, pause_every=n // 5, pause_len=4000, dtype=dtype)
    f = str(tmp_path / ("capture" + ext))
    iq.tofile(f)
    scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64) - (128 if dtype == np.uint8 else 0)).max())
    par = dict(modulation_type="FSK", samples_per_symbol=100, center=0.0, tolerance=5, noise_threshold=0.1 * scale, pause_threshold=8)
    s = Signal.from_file_streamed(f, **par)
    ref = Signal.from_file(f)
    for k, v in par.items():
        setattr(ref, k, v)
    assert np.array_equal(s.iq.cpu().numpy(), ref.iq.cpu().numpy())
    upload_passes = s.demod_passes
    assert np.array_equal(s.qad.cpu().numpy().view(np.uint32), ref.qad.cpu().numpy().view(np.uint32))
    passes = s.demod_passes
    assert np.array_equal(s.ppseq(), ref.ppseq())
    for a, b in zip(s._digitize()[1:], ref._digitize()[1:]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    if dtype != np.uint8:
        assert s.demod_passes == upload_passes == 1               # nothing was demodulated again: the upload pass left everything in place
    host = ref.iq.cpu().numpy()
    qad = oracle.afp_demod(host, par["noise_threshold"], "FSK", 2)
    pp = oracle.grab_pulse_lens(qad, 0.0, 5, "FSK", 100, 1, 1.0)
    assert np.array_equal(s.qad.cpu().numpy().view(np.uint32), qad.view(np.uint32)) and np.array_equal(s.ppseq(), pp)
    flat = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
    for a, b in zip(s._digitize()[1:], flat):
        assert np.array_equal(np.asarray(a), b)
    s.center = 0.05                                               # a slicing parameter: re-sliced from the cached qad, no new pass over the samples
    pp2 = oracle.grab_pulse_lens(qad, 0.05, 5, "FSK", 100, 1, 1.0)
    assert np.array_equal(s.ppseq(), pp2) and s.demod_passes == passes


@pytest.mark.gpu
def test_from_file_streamed_keeps_its_stream_between_files(oracle, tmp_path):
    """`pinned=` keeps the pinned read buffer and the capture stream between calls (a stream -- three output slots, six pinned
    blobs -- was built and destroyed per file): three captures of one size class opened one after the other through ONE stream, each equal
    to the reference's digitisation; another parameter set rebuilds the stream; the earlier Signals stay valid (their results were copied)."""
    from urh_amd.signal import Signal
    n = (1 << 20) + 4096
    par = dict(modulation_type="FSK", samples_per_symbol=100, center=0.0, tolerance=5, noise_threshold=0.1, pause_threshold=8)
    keep, sigs, files = {}, [], []
    for i in range(3):
        iq = synth_fsk(n - 8192 * i, sps=100, seed=70 + i, noise=0.04, pause_every=n // 4, pause_len=3000 + 500 * i)
        f = str(tmp_path / f"c{i}.complex")
        iq.tofile(f)
        files.append(iq)
        sigs.append(Signal.from_file_streamed(f, pinned=keep, **par))
        if i == 0:
            first_stream = keep["stream"]
        assert keep["stream"] is first_stream and first_stream.stats()["pushed"] == i + 1
    for iq, s in zip(files, sigs):
        qad = oracle.afp_demod(iq, 0.1, "FSK", 2)
        pp = oracle.grab_pulse_lens(qad, 0.0, 5, "FSK", 100, 1, 1.0)
        assert np.array_equal(s.qad.cpu().numpy().view(np.uint32), qad.view(np.uint32)) and np.array_equal(s.ppseq(), pp)
        for a, b in zip(s._digitize()[1:], oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)):
            assert np.array_equal(np.asarray(a), b)
    s4 = Signal.from_file_streamed(str(tmp_path / "c0.complex"), pinned=keep, **dict(par, tolerance=3))
    assert keep["stream"] is not first_stream
    pp = oracle.grab_pulse_lens(oracle.afp_demod(files[0], 0.1, "FSK", 2), 0.0, 3, "FSK", 100, 1, 1.0)
    assert np.array_equal(s4.ppseq(), pp)
    keep["stream"].close()


@pytest.mark.gpu
def test_from_file_streamed_falls_back_when_the_stream_cannot_take_the_capture(oracle, tmp_path):
    """A noise-dominated capture read with tolerance 0 and no noise gate has far more pulse-table rows than the stream's default capacity
    (about four per symbol): the streamed route reports `truncated` and the Signal falls back to the ordinary passes with their capacity
    retry on the capture that is by then resident -- the same qad, pulse table and bits as from_file, equal to the oracle.  Slicing
    parameters the stream rejects (samples_per_symbol = 0) leave a Signal that raises where the reference raises, in bits()."""
    from urh_amd.signal import Signal
    n = 1 << 22
    iq = (0.3 * np.random.default_rng(8).standard_normal((n, 2))).astype(np.float32)
    f = str(tmp_path / "noise.complex")
    iq.tofile(f)
    par = dict(modulation_type="FSK", samples_per_symbol=100, center=0.0, tolerance=0, noise_threshold=0.0, pause_threshold=8)
    s = Signal.from_file_streamed(f, **par)
    assert np.array_equal(s.iq.cpu().numpy(), iq)
    qad = oracle.afp_demod(iq, 0.0, "FSK", 2)
    pp = oracle.grab_pulse_lens(qad, 0.0, 0, "FSK", 100, 1, 1.0)
    assert len(pp) > 4 * (n // 100) + 4096                          # (beyond the stream's default capacity: the fallback was needed)
    assert np.array_equal(s.qad.cpu().numpy().view(np.uint32), qad.view(np.uint32))
    assert np.array_equal(s.ppseq(), pp)
    flat = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
    for a, b in zip(s._digitize()[1:], flat):
        assert np.array_equal(np.asarray(a), b)
    s2 = Signal.from_file_streamed(f, **dict(par, samples_per_symbol=0, tolerance=5))
    assert np.array_equal(s2.iq.cpu().numpy(), iq)
    assert np.array_equal(s2.qad.cpu().numpy().view(np.uint32), qad.view(np.uint32))
    with pytest.raises(ZeroDivisionError):
        s2.bits()


@pytest.mark.gpu
def test_two_live_results_do_not_share_a_stale_host_cache(oracle):
    """BitsResult.host() views a pinned buffer of the pipeline that every result's host() overwrites: r0.ppseq(); r1.ppseq(); r0.flat() must
    give r0's bits, not r1's blob read with r0's offsets."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    pipe = DevicePipeline(0)
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    caps = [synth_fsk(1 << 20, sps=100, seed=60 + i, noise=0.04, pause_every=(1 << 20) // (3 + 2 * i), pause_len=3000 + 500 * i) for i in range(2)]
    want = []
    for iq in caps:
        qad = oracle.afp_demod(iq, 0.1, "FSK", 2)
        pp = oracle.grab_pulse_lens(qad, 0.0, 5, "FSK", 100, 1, 1.0)
        want.append((pp, oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)))
    r = [pipe.iq_to_bits(torch.from_numpy(iq).cuda(), p, want_qad=False, slot=i) for i, iq in enumerate(caps)]
    assert np.array_equal(r[0].ppseq(), want[0][0])
    assert np.array_equal(r[1].ppseq(), want[1][0])
    for a, b in zip(r[0].flat(), want[0][1]):
        assert np.array_equal(a, b)
    for a, b in zip(r[1].flat(), want[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(r[0].ppseq(), want[0][0])


def test_dialog_helpers_against_the_real_signal(pipe):
    """calc_relative_noise_threshold_from_range / get_thresholds_for_center / center_thresholds / create_new / from_samples on the real
    reference Signal and on the shim: the same numbers (Signal.py:486-535, :659-664)"""
    import ref_python
    if not ref_python.available():
        pytest.skip("oracle/_ref (compiled reference + Python sources) not present")
    ref_python.setup()
    from urh.signalprocessing.IQArray import IQArray
    from urh.signalprocessing.Signal import Signal as RefSignal
    from urh_amd.signal import Signal
    rng = np.random.default_rng(77)
    for dtype in (np.float32, np.int8, np.uint8, np.int16, np.uint16):
        iq = synth_fsk(50_000, sps=80, seed=9, noise=0.05, pause_every=12_000, pause_len=3000, dtype=dtype)
        ref = RefSignal.from_samples(iq, "x", 2e6)
        mine = Signal.from_samples(iq, "x", 2e6, pipe=pipe)
        assert mine.sample_rate == 2e6 and mine.name == "x"
        for a, b in [(0, 50_000), (12_001, 14_999), (14_999, 12_001), (7, 8), (33_333, 33_334), (100, 100), (-3000, -1), (49_000, 60_000)] + \
                    [tuple(int(v) for v in rng.integers(0, 50_000, 2)) for _ in range(10)]:
            want = ref.calc_relative_noise_threshold_from_range(a, b)
            got = mine.calc_relative_noise_threshold_from_range(a, b)
            assert float(got) == float(want) or (np.isnan(got) and np.isnan(want)), (np.dtype(dtype).name, a, b, got, want)    # (uint16: NaN in both)
        for mod_bits, center, spacing in ((1, 0.02, 1.0), (2, -0.1, 0.3), (3, 0.0, 0.05)):
            ref.bits_per_symbol = mine.bits_per_symbol = mod_bits
            ref.center = mine.center = center
            ref.center_spacing = mine.center_spacing = spacing
            assert np.array_equal(np.asarray(ref.center_thresholds, np.float32).view(np.uint32), np.asarray(mine.center_thresholds, np.float32).view(np.uint32))
            assert np.array_equal(np.asarray(ref.get_thresholds_for_center(0.3, 0.2), np.float32).view(np.uint32),
                                  np.asarray(mine.get_thresholds_for_center(0.3, 0.2), np.float32).view(np.uint32))
        ref.noise_threshold = mine.noise_threshold = 3.0 if dtype != np.float32 else 0.03
        ref.samples_per_symbol = mine.samples_per_symbol = 80
        rn, mn = ref.create_new(1000, 9000), mine.create_new(1000, 9000)
        assert np.array_equal(mn.iq.cpu().numpy(), rn.iq_array.data) and mn.name == rn.name and mn.changed and rn.changed
        assert (mn.noise_threshold, mn.samples_per_symbol, mn.bits_per_symbol, mn.center, mn.sample_rate, mn.timestamp) == \
               (rn.noise_threshold, rn.samples_per_symbol, rn.bits_per_symbol, rn.center, rn.sample_rate, rn.timestamp)
        other = synth_fsk(777, sps=10, seed=3, dtype=dtype)
        rn, mn = ref.create_new(new_data=other, new_timestamp=4.5), mine.create_new(new_data=other, new_timestamp=4.5)
        assert np.array_equal(mn.iq.cpu().numpy(), rn.iq_array.data) and mn.timestamp == rn.timestamp == 4.5
    mine.eliminate()
    assert mine.iq is None


def _np_estimate_frequency(iq_c64, start, end, sample_rate):
    """Signal.estimate_frequency (Signal.py:578-601), the reference's lines on a complex64 array"""
    import math
    length = 2 ** int(math.log2(end - start))
    data = iq_c64[start:start + length]
    try:
        w = np.fft.fft(data)
        frequencies = np.fft.fftfreq(len(w))
        idx = np.argmax(np.abs(w))
        return abs(frequencies[idx] * sample_rate), w, int(idx)
    except ValueError:
        return 100e3, None, None


def test_estimate_frequency_equals_numpy(pipe):
    """tones (one dominant carrier + a weaker one + noise) over windows of 2 .. 2^24 samples -- one LDS-sized transform up to 8192, the
    four-step form beyond --, float32 and int16 captures: the frequency numpy's FFT gives (Signal.py:578-601).  Pure noise: the bin found
    holds a magnitude within float32 rounding of numpy's largest."""
    from urh_amd.signal import Signal
    rng = np.random.default_rng(31)
    n_all = (1 << 24) + 12345
    t = np.arange(n_all, dtype=np.float64)
    for case, (f0, dtype) in enumerate([(0.0371, np.float32), (-0.2113, np.float32), (0.4983, np.int16), (0.00002, np.float32)]):
        x = np.exp(2j * np.pi * f0 * t) + 0.3 * np.exp(2j * np.pi * (f0 / 3 + 0.11) * t)
        x = x + 0.2 * (rng.standard_normal(n_all) + 1j * rng.standard_normal(n_all))
        iq = np.stack([x.real, x.imag], 1)
        if dtype == np.float32:
            iq = (iq * 0.5).astype(np.float32)
            c64 = iq.view(np.complex64).reshape(-1)
        else:
            iq = np.clip(np.round(iq * 8000), -32768, 32767).astype(np.int16)
            c64 = (iq.astype(np.float32) / np.float32(32768.0)).view(np.complex64).reshape(-1)      # IQArray.as_complex64 (:92-93, :171-181)
        sig = Signal(iq, pipe=pipe)
        for log2len in [1, 2, 3, 5, 8, 12, 13, 14, 15, 17, 20, 22] + ([24] if case == 0 else []):
            if n_all > (2 << log2len):
                start = int(rng.integers(0, n_all - (2 << log2len)))
                end = start + (1 << log2len) + int(rng.integers(0, 1 << log2len))   # the window is the largest power of two that fits
            else:
                start, end = 77, n_all
            want, w, idx = _np_estimate_frequency(c64, start, end, 2e6)
            got = sig.estimate_frequency(start, end, 2e6)
            if got != want:                                  # a near-tie between bins (short windows): the magnitudes must agree to rounding
                mags = np.abs(w)
                k = int(round(got / 2e6 * len(w)))
                cand = [mags[k % len(w)], mags[(-k) % len(w)]]
                assert max(cand) >= mags[idx] * (1 - 2e-5), (case, log2len, start, got, want)
            if log2len >= 12:
                assert got == want, (case, log2len, start, got, want)
    assert sig.estimate_frequency(100, 100, 1e6) == 100e3 and sig.estimate_frequency(200, 100, 1e6) == 100e3
    noise = (rng.standard_normal((1 << 16, 2))).astype(np.float32)
    sig = Signal(noise, pipe=pipe)
    want, w, idx = _np_estimate_frequency(noise.view(np.complex64).reshape(-1), 0, 1 << 16, 1.0)
    got = sig.estimate_frequency(0, 1 << 16, 1.0)
    k = int(round(got * (1 << 16)))
    assert max(np.abs(w)[k], np.abs(w)[-k]) >= np.abs(w)[idx] * (1 - 2e-5)
