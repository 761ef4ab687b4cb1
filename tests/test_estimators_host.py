"""Host-side decision logic (no GPU involved) against the reference's own functions (/root/reference: AutoInterpretation.py,
auto_interpretation.pyx) on randomised plateau lengths / segments: the library's native per-message decisions
(urhgpu_msg_bit_lengths and the two halves around np.argsort), the array-form helpers of urh_amd/estimators.py, and the numpy
restatements the gpu tests use as their comparison (tests/numpy_estimators.py).  Needs the oracle/_ref build; skipped where
/root/reference is absent."""
import numpy as np
import pytest

import numpy_estimators as ne
from urh_amd import estimators as e


@pytest.fixture(scope="module")
def ref():
    import build_ref
    import ref_python
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("reference Python not available")
    ref_python.setup()
    from urh.ainterpretation import AutoInterpretation as AI
    from urh.cythonext import auto_interpretation as c_ai
    return AI, c_ai


def _plateaus(rng):
    n = int(rng.integers(0, 400))
    base = int(rng.choice([8, 40, 100, 295]))
    p = base * rng.integers(1, 6, n) + rng.integers(-3, 4, n)
    glitch = rng.random(n) < rng.choice([0.0, 0.1, 0.3])
    p[glitch] = rng.integers(1, 4, int(glitch.sum()))
    return np.maximum(p, 1).astype(np.uint64)


def test_plateau_logic_equals_reference(ref):
    AI, c_ai = ref
    rng = np.random.default_rng(0)
    for it in range(400):
        p = _plateaus(rng)
        tol_ref = AI.estimate_tolerance_from_plateau_lengths(p)
        tol = ne.estimate_tolerance_from_plateau_lengths(p)
        assert (tol_ref is None and tol is None) or int(tol_ref) == int(tol), (it, tol_ref, tol)
        for t in (0, 1, 3, 5, None):
            a = AI.merge_plateau_lengths(p.copy(), tolerance=t)
            b = ne.merge_plateau_lengths(p.copy(), tolerance=t)
            assert np.array_equal(np.asarray(a), np.asarray(b)), (it, t)
        if len(p):
            assert np.array_equal(np.asarray(c_ai.get_threshold_divisor_histogram(p.copy())), ne.get_threshold_divisor_histogram(p.copy())), it
            assert np.array_equal(np.asarray(c_ai.merge_plateaus(p.copy(), 2, 50)), e.merge_plateaus(p.copy(), 2, 50)), it
        a, b = p.copy(), p.copy()
        assert AI.get_bit_length_from_plateau_lengths(a) == ne.get_bit_length_from_plateau_lengths(b), it
        assert np.array_equal(a, b)                              # rounded in place the same way
        # the product: the per-message chain of AutoInterpretation.estimate (:416-433) natively, ties through np.argsort
        merged = AI.merge_plateau_lengths(p.copy(), tolerance=0 if tol_ref is None else tol_ref)
        want_len = AI.get_bit_length_from_plateau_lengths(np.array(merged, dtype=np.uint64)) if len(merged) >= 2 else None
        got_tol, got_len = e.bit_length_of_message(p.copy())
        assert (tol_ref is None and got_tol is None) or int(tol_ref) == int(got_tol), (it, tol_ref, got_tol)
        assert (want_len is None and got_len is None) or int(want_len) == int(got_len), (it, want_len, got_len)
        got_tol2, got_len2 = e._bit_length_with_numpy_order(p.copy())      # the tie path gives the same on every message
        assert (got_tol2, got_len2) == (got_tol, got_len), it


def test_multiset_decisions_equal_the_sequence_decisions():
    """urhgpu_msg_plateau_decisions decides a message from the (value, count) pairs of its plateau lengths where the order does not matter
    (tolerance <= 0): the multiset form (host hook) gives what urhgpu_msg_bit_lengths gives on the sequence, and asks for the sequence
    exactly where the tolerance is positive.  No reference needed: two implementations of this library against each other."""
    import ctypes as C
    from urh_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    seen = {"counts": 0, "sequence": 0, "ties": 0}
    for it in range(3000):
        if it % 3 == 0:                                    # clean messages: multiples of a symbol length with jitter, thousands of plateaus
            n = int(rng.integers(0, 6000))
            base = int(rng.choice([8, 25, 100, 295, 1000, 12345]))
            p = base * rng.integers(1, 5, n) + rng.integers(-2, 3, n) * int(rng.integers(0, 2))
            p = np.maximum(p, 1).astype(np.uint64)
        else:
            p = _plateaus(rng)
        if it % 50 == 1:
            p = p[:int(rng.integers(0, 3))]
        off = np.array([0, len(p)], dtype=np.int64)
        tol, bl = np.zeros(1, np.int64), np.zeros(1, np.int64)
        buf = p if len(p) else np.zeros(1, np.uint64)
        _lib.check(lib.urhgpu_msg_bit_lengths(buf.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 1, tol.ctypes.data_as(C.c_void_p),
                                              bl.ctypes.data_as(C.c_void_p)))
        t2, b2 = C.c_int64(0), C.c_int64(0)
        _lib.check(lib.urhgpu_test_bit_length_from_counts(buf.ctypes.data_as(C.c_void_p), len(p), C.byref(t2), C.byref(b2)))
        if t2.value == -3:
            assert b2.value == -3 and tol[0] > 0, (it, tol[0], bl[0])
            seen["sequence"] += 1
        else:
            assert (t2.value, b2.value) == (int(tol[0]), int(bl[0])), (it, t2.value, b2.value, tol[0], bl[0], len(p))
            seen["counts"] += 1
            seen["ties"] += int(bl[0] == -2)
    assert seen["counts"] > 800 and seen["sequence"] > 300, seen


def test_peaks_center_equals_reference_walk(ref):
    """estimators.peaks_center (array form, used where np.argsort has to break a tie) against the reference's loop over
    np.argsort(y)[::-1] (AutoInterpretation.py:250-277, restated in numpy_estimators.center_from_histogram)"""
    rng = np.random.default_rng(5)
    for it in range(500):
        nb = int(rng.integers(1, 120))
        y = rng.integers(0, int(rng.choice([2, 5, 1000])), nb)
        edges = np.cumsum(rng.random(nb + 1))
        want = ne.center_from_histogram(y, edges)
        got = e.peaks_center(y, edges)
        assert (want is None and got is None) or float(want) == float(got), (it, y.tolist())


def test_segment_and_value_helpers_equal_reference(ref):
    AI, _ = ref
    rng = np.random.default_rng(1)
    for it in range(300):
        n = int(rng.integers(0, 30))
        starts = np.cumsum(rng.integers(5, 4000, n))
        segs = [(int(s), int(s + rng.integers(1, 3000))) for s in starts]
        segs = [s for i, s in enumerate(segs) if i == 0 or s[0] > segs[i - 1][1]] if segs else segs
        segs2 = []
        for s in segs:
            if not segs2 or s[0] > segs2[-1][1]:
                segs2.append(s)
        a = AI.merge_message_segments_for_ook(list(segs2))
        b = e.merge_message_segments_for_ook(list(segs2))
        assert [tuple(map(int, x)) for x in a] == [tuple(map(int, x)) for x in b], (it, segs2)
        from oracle import urh_oracle
        c = urh_oracle.merge_message_segments_for_ook(list(segs2))
        assert [tuple(map(int, x)) for x in a] == [tuple(map(int, x)) for x in c], (it, segs2)
        vals = [int(v) for v in rng.integers(0, 6, int(rng.integers(0, 12)))]
        assert AI.get_most_frequent_value(list(vals)) == e.get_most_frequent_value(list(vals)), vals
        d = rng.standard_normal(int(rng.integers(0, 20)))
        for fn, red, z in (("max_without_outliers", np.max, 3), ("min_without_outliers", np.min, 2)):
            x = getattr(AI, fn)(d)
            inl = e._inliers(d, z)
            assert (x is None and len(inl) == 0) or x == red(inl)


def test_median_filter_equals_reference(ref):
    AI, c_ai = ref
    rng = np.random.default_rng(4)
    for n in (0, 1, 2, 5, 10, 11, 12, 100, 1000):
        x = rng.standard_normal(n) * rng.choice([1e-3, 1.0, 1e6])
        for k in (1, 2, 3, 11, 12):
            want = c_ai.median_filter(x, k=k)
            got = ne.median_filter(x, k=k)
            assert got.dtype == want.dtype and np.array_equal(got, want), (n, k)


def test_detect_modulation_equals_reference(ref):
    """AutoInterpretation.detect_modulation on every message the reference segments out of the golden captures, plus
    synthetic ASK / PSK / noise messages: the same label."""
    import os
    from conftest import GOLDEN_DIR
    AI, c_ai = ref
    from urh.signalprocessing.IQArray import IQArray
    n_checked = 0
    for f in sorted(os.listdir(GOLDEN_DIR)):
        if not f.endswith(".npz"):
            continue
        z = np.load(os.path.join(GOLDEN_DIR, f))
        if "iq" not in z:
            continue
        iq = z["iq"]
        arr = IQArray(iq)
        noise = AI.detect_noise_level(arr.magnitudes)
        segs = AI.segment_messages_from_magnitudes(arr.magnitudes, noise_threshold=noise)
        cplx = arr.as_complex64()
        assert np.array_equal(ne._as_complex64(iq).view(np.uint32), cplx.view(np.uint32)), f
        for start, end in segs[:100]:
            assert ne.detect_modulation(cplx[start:end]) == AI.detect_modulation(cplx[start:end]), (f, start, end)
            n_checked += 1
    rng = np.random.default_rng(12)
    t = np.arange(4096)
    for msg in (np.exp(2j * np.pi * 0.05 * t) * np.repeat(rng.integers(1, 3, 64), 64),
                np.exp(2j * np.pi * 0.05 * t + 1j * np.pi * np.repeat(rng.integers(0, 2, 64), 64)),
                rng.standard_normal(4096) + 1j * rng.standard_normal(4096), np.zeros(100), np.ones(3)):
        msg = (msg + 0.01 * (rng.standard_normal(len(msg)) + 1j * rng.standard_normal(len(msg))) * (np.abs(msg) > 0)).astype(np.complex64)
        assert ne.detect_modulation(msg) == AI.detect_modulation(msg)
        n_checked += 1
    assert n_checked > 20


def test_positions_derived_from_the_pulse_table_equal_the_oracle(oracle):
    """urh_amd.pipeline.positions_from_rows (what HostBits.bit_sample_pos() gives when a capture stream ships no positions): the
    bit_sample_pos arrays of ProtocolAnalyzer._ppseq_to_bits (:346-411) from the pulse table alone, on random tables (groups without data,
    leading pauses, long pauses with and without a message before them, every pause_threshold rule) and on the golden captures."""
    from conftest import GOLDEN_CASES, load_golden
    from urh_amd.pipeline import DemodParams, positions_from_rows
    rng = np.random.default_rng(4)
    for it in range(600):
        n = int(rng.integers(0, 80))
        sps = int(rng.choice([1, 7, 100]))
        bps = int(rng.choice([1, 2, 3]))
        pt = int(rng.choice([0, 1, 8]))
        pp = np.stack([rng.integers(-1, 2 ** bps, n), rng.integers(1, sps * 14 + 3, n)], 1).astype(np.int64).reshape(-1, 2)
        want = oracle.ppseq_to_bits_flat(pp, sps, bps, True, pt)
        pos, off = positions_from_rows(pp[:, 0], pp[:, 1], DemodParams("FSK", bps, 0, 0, 1, 5, sps, 0.1, pt, False))
        assert np.array_equal(pos, want[3]) and np.array_equal(off, want[4]), (it, pp.tolist())
    for name in GOLDEN_CASES:
        g = load_golden(name)
        if "ppseq" not in g or g["modulation_type"] == "PSK":
            continue
        pp = g["ppseq"]
        want = oracle.ppseq_to_bits_flat(pp, g["samples_per_symbol"], g["bits_per_symbol"], True, g["pause_threshold"])
        pos, off = positions_from_rows(pp[:, 0], pp[:, 1], DemodParams(g["modulation_type"], g["bits_per_symbol"], 0, 0, 1, 5, g["samples_per_symbol"], 0.1,
                                                                       g["pause_threshold"], False))
        assert np.array_equal(pos, want[3]) and np.array_equal(off, want[4]), name


def test_host_bits_from_a_blob_in_host_memory():
    """pipeline.HostBits.from_blob: the compact result blob (include/urhgpu.h) parsed from plain host memory -- the layout restated here
    byte for byte: header of 16 int64 with the section offsets, 16-byte aligned sections"""
    import ctypes as C
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, HostBits
    rng = np.random.default_rng(5)
    n_rows, n_msg = 37, 3
    row_state = rng.integers(-1, 2, n_rows).astype(np.int8)
    row_len = rng.integers(1, 5000, n_rows).astype(np.int32)
    bits = rng.integers(0, 2, 101).astype(np.uint8)
    msg_off = np.array([0, 40, 77, 101], np.int64)
    pauses = np.array([300, 0, 12345], np.int64)
    pos = np.sort(rng.integers(0, 1 << 31, 101 + 2 * n_msg)).astype(np.uint32)
    pos_off = np.array([0, 42, 81, 107], np.int64)
    for has_pos in (1, 0):
        sections, off = {}, 128

        def put(name, arr):
            nonlocal off
            off = (off + 15) & ~15
            sections[name] = (off, np.ascontiguousarray(arr).tobytes())
            off += len(sections[name][1])
        put("pauses", pauses); put("msg_off", msg_off); put("pos_off", pos_off); put("row_state", row_state)
        put("bits", np.packbits(bits)); put("row_len", row_len)
        if has_pos:
            put("pos32", pos)
        total = off
        hdr = np.zeros(16, np.int64)
        hdr[:8] = [_lib.BLOB_MAGIC, n_rows, n_msg, len(bits), len(pos) if has_pos else 0, n_rows, total, has_pos]
        hdr[8:15] = [sections["pauses"][0], sections["msg_off"][0], sections["pos_off"][0], sections["row_state"][0], sections["bits"][0],
                     sections["row_len"][0], sections["pos32"][0] if has_pos else 0]
        blob = bytearray(total)
        blob[:128] = hdr.tobytes()
        for o, b in sections.values():
            blob[o:o + len(b)] = b
        buf = (C.c_ubyte * total).from_buffer(blob)
        h = HostBits.from_blob(C.addressof(buf), DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, bool(has_pos)), seq=7, n_samples=1 << 20).check()
        assert (h.seq, h.n_rows, h.n_msg, h.n_bits, h.blob_bytes, h.truncated) == (7, n_rows, n_msg, len(bits), total, False)
        assert np.array_equal(h.ppseq(), np.stack([row_state.astype(np.int64), row_len.astype(np.int64)], axis=1))
        assert np.array_equal(h.bits(), bits) and np.array_equal(h.pauses, pauses) and np.array_equal(h.msg_off, msg_off)
        if has_pos:
            assert np.array_equal(h.bit_sample_pos(), pos.astype(np.int64)) and np.array_equal(h.pos_offsets(), pos_off)
        else:
            h.sharded_piece = True                           # a rank's piece of a sharded capture: positions must have been shipped
            with pytest.raises(ValueError):
                h.bit_sample_pos()
    with pytest.raises(ValueError):
        HostBits.from_blob(C.addressof((C.c_ubyte * 128)()), None)


def test_host_bits_from_a_blob_with_16_bit_row_lengths():
    """The blob of a staged pass (include/urhgpu.h: URHGPU_BLOB_LEN16, header[7] bit 1): row_len as uint16, rows of 65535 samples and more -- and a
    negative length (signal_functions.pyx:485-493: the last row of a capture shorter than the tolerance) -- as 0xFFFF with {uint32 row, int32 length}
    entries in the escape list behind the packed bits.  HostBits widens; a 0xFFFF without its entry is an error, not a length."""
    import ctypes as C
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, HostBits
    rng = np.random.default_rng(11)
    n_rows = 53
    row_state = rng.integers(-1, 2, n_rows).astype(np.int8)
    row_len = rng.integers(1, 60000, n_rows).astype(np.int64)
    row_len[[3, 17, 52]] = [65535, 1 << 30, -4]                 # the boundary, a long pause, a negative last row
    row_len[9] = 65534                                           # the largest length that ships as it is
    bits = rng.integers(0, 2, 77).astype(np.uint8)
    msg_off, pauses, pos_off = np.array([0, 77], np.int64), np.array([0], np.int64), np.array([0, 78], np.int64)
    for drop_entry in (False, True):
        esc_rows = [r for r in range(n_rows) if not 0 <= row_len[r] < 0xFFFF]
        assert esc_rows == [3, 17, 52]
        len16 = np.where((row_len >= 0) & (row_len < 0xFFFF), row_len, 0xFFFF).astype(np.uint16)
        listed = esc_rows[:-1] if drop_entry else esc_rows
        esc = np.zeros(1 + len(listed), np.int64)
        esc[0] = len(listed)
        pairs = np.zeros((len(listed), 2), np.uint32)
        pairs[:, 0] = listed
        pairs[:, 1] = row_len[listed].astype(np.int32).view(np.uint32)
        sections, off = {}, 128

        def put(name, raw):
            nonlocal off
            off = (off + 15) & ~15
            sections[name] = (off, raw)
            off += len(raw)
        put("pauses", pauses.tobytes()); put("msg_off", msg_off.tobytes()); put("pos_off", pos_off.tobytes())
        put("bits", np.packbits(bits).tobytes())
        put("esc", esc[:1].tobytes() + pairs.tobytes())              # behind the packed bits, at the next 16-byte boundary
        put("row_state", row_state.tobytes()); put("row_len", len16.tobytes())
        total = off
        hdr = np.zeros(16, np.int64)
        hdr[:8] = [_lib.BLOB_MAGIC, n_rows, 1, len(bits), 0, n_rows, total, _lib.BLOB_LEN16]
        hdr[8:15] = [sections["pauses"][0], sections["msg_off"][0], sections["pos_off"][0], sections["row_state"][0], sections["bits"][0],
                     sections["row_len"][0], 0]
        blob = bytearray(total)
        blob[:128] = hdr.tobytes()
        for o, b in sections.values():
            blob[o:o + len(b)] = b
        buf = (C.c_ubyte * total).from_buffer(blob)
        h = HostBits.from_blob(C.addressof(buf), DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False), n_samples=1 << 20).check()
        if drop_entry:
            with pytest.raises(_lib.UrhGpuError):
                h.ppseq()
        else:
            assert h.row_len.dtype == np.int32
            assert np.array_equal(h.ppseq(), np.stack([row_state.astype(np.int64), row_len], axis=1))
            assert np.array_equal(h.bits(), bits)


def test_host_bits_from_a_blob_with_packed_rows():
    """The blob of a staged pass over a dense pulse table (include/urhgpu.h: URHGPU_BLOB_ROW16, header[7] bit 2): no row_state section, one uint16
    per row = (state + 1) << 13 | length, lengths of 8191 samples and more (and a negative one) as 0x1FFF with their entries in the escape list
    at header[11].  HostBits gives row_state and row_len back; a 0x1FFF without its entry is an error."""
    import ctypes as C
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, HostBits
    rng = np.random.default_rng(12)
    n_rows = 61
    row_state = rng.integers(-1, 4, n_rows).astype(np.int8)          # pause, and the four states of an order-4 capture
    row_len = rng.integers(1, 8000, n_rows).astype(np.int64)
    row_len[[2, 30, 60]] = [8191, 1 << 29, -7]                    # the boundary, a long pause, a negative last row
    row_len[5] = 8190                                             # the largest length that ships as it is
    bits = rng.integers(0, 2, 90).astype(np.uint8)
    msg_off, pauses, pos_off = np.array([0, 90], np.int64), np.array([0], np.int64), np.array([0, 91], np.int64)
    for drop_entry in (False, True):
        esc_rows = [r for r in range(n_rows) if not 0 <= row_len[r] < 0x1FFF]
        assert esc_rows == [2, 30, 60]
        words = (((row_state.astype(np.int64) + 1) << 13) | np.where((row_len >= 0) & (row_len < 0x1FFF), row_len, 0x1FFF)).astype(np.uint16)
        listed = esc_rows[1:] if drop_entry else esc_rows
        pairs = np.zeros((len(listed), 2), np.uint32)
        pairs[:, 0] = listed
        pairs[:, 1] = row_len[listed].astype(np.int32).view(np.uint32)
        sections, off = {}, 128

        def put(name, raw):
            nonlocal off
            off = (off + 15) & ~15
            sections[name] = (off, raw)
            off += len(raw)
        put("pauses", pauses.tobytes()); put("msg_off", msg_off.tobytes()); put("pos_off", pos_off.tobytes())
        put("bits", np.packbits(bits).tobytes())
        put("row_len", words.tobytes())
        put("esc", np.array([len(listed)], np.int64).tobytes() + pairs.tobytes())        # a place of its own behind the sections
        total = off
        hdr = np.zeros(16, np.int64)
        hdr[:8] = [_lib.BLOB_MAGIC, n_rows, 1, len(bits), 0, n_rows, total, _lib.BLOB_ROW16]
        hdr[8:15] = [sections["pauses"][0], sections["msg_off"][0], sections["pos_off"][0], sections["esc"][0], sections["bits"][0],
                     sections["row_len"][0], 0]
        blob = bytearray(total)
        blob[:128] = hdr.tobytes()
        for o, b in sections.values():
            blob[o:o + len(b)] = b
        buf = (C.c_ubyte * total).from_buffer(blob)
        h = HostBits.from_blob(C.addressof(buf), DemodParams("FSK", 2, 0.0, 0.0, 0.1, 1, 10, 0.1, 8, False), n_samples=1 << 20).check()
        if drop_entry:
            with pytest.raises(_lib.UrhGpuError):
                h.ppseq()
        else:
            assert h.row_len.dtype == np.int32 and h.row_state.dtype == np.int8
            assert np.array_equal(h.row_state, row_state)
            assert np.array_equal(h.ppseq(), np.stack([row_state.astype(np.int64), row_len], axis=1))
            assert np.array_equal(h.bits(), bits)


def test_peaks_center_linear_form_equals_the_offset_loop():
    """peaks_center finds the strict maxima with two sliding-window maxima (linear in the bins: a nearly constant message has millions);
    the per-offset comparison loop it replaced is the checker here (AutoInterpretation.py:250-277)"""
    from urh_amd.estimators import peaks_center

    def by_offsets(counts, edges):
        y = np.asarray(counts, dtype=np.int64)
        nb = len(y)
        reach = max(2, int(0.05 * nb) + 1) - 1
        padded = np.concatenate([np.zeros(reach, np.int64), y, np.zeros(reach, np.int64)])
        peak = y > 0
        for d in range(1, reach + 1):
            peak &= (y > padded[reach - d:reach - d + nb]) & (y > padded[reach + d:reach + d + nb])
        if not peak.any():
            return None
        walk = np.argsort(counts)[::-1]
        return np.mean(np.asarray(edges)[walk[peak[walk]][:2]])
    rng = np.random.default_rng(3)
    for it in range(1500):
        nb = int(rng.choice([1, 2, 3, 5, 19, 20, 21, 40, 41, 100, 777, int(rng.integers(1, 3000))]))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            y = rng.integers(0, 5, nb)
        elif kind == 1:
            y = np.zeros(nb, np.int64)
            y[rng.integers(0, nb, max(1, nb // 20))] += rng.integers(1, 4, max(1, nb // 20))
        elif kind == 2:
            y = rng.integers(0, 1000, nb)
        else:
            y = np.full(nb, int(rng.integers(0, 3)))
        e = np.arange(nb + 1) * 0.37 - 5
        a, b = by_offsets(y, e), peaks_center(y, e)
        assert (a is None and b is None) or a == b, (it, nb, a, b)
    import time
    y = np.zeros(2_000_000, np.int64)
    y[rng.integers(0, len(y), 80)] += 1
    y[1000], y[1_500_000] = 9, 7
    t0 = time.perf_counter()
    assert peaks_center(y, np.arange(len(y) + 1) * 1e-9) == np.mean([1000e-9, 1_500_000e-9])
    assert time.perf_counter() - t0 < 20.0
