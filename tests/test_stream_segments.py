"""Streamed passes (urhgpu_stream_*, pulse_table.hip "Segments"): the tail of a capture runs in SEGMENTS beside the hot kernel, every
segment storing its share of the compact blob straight into pinned host memory.  Whatever the segmentation -- number of segments, equal
or halving, the open message group a boundary cuts through, groups without a data row that are dropped after all -- the host receives
exactly what the reference computes for the capture (oracle = the C restatement pinned on the reference, tests/test_oracle.py).

Captures are cut into chunks of ONE tile here (urhgpu_test_force_tiles_per_chunk), so that 2^22 samples are 2048 chunks = eight
segment-alignment units: every boundary a 1 GiB capture's segmentation can have is exercised at a size the oracle finishes in a second."""
import ctypes as C

import numpy as np
import pytest

from conftest import synth_fsk

pytestmark = pytest.mark.gpu

N = 1 << 22
UNIT = 256 * 2048          # samples per segment-alignment unit with one tile per chunk


def _events_capture(n, seed, boundary_trains=True):
    """2-FSK bursts under an on/off envelope: long data bursts, long pauses (close a message), short pauses (zero bits) and TINY bursts
    (longer than the tolerance, shorter than half a symbol: a data row of zero symbols -- a message group that holds only those and
    short pauses has no data and is dropped when its long pause comes).  boundary_trains: trains of tiny bursts and short pauses
    across every multiple of UNIT, some ended by a data burst (kept), some by a long pause (dropped)."""
    rng = np.random.default_rng(seed)
    env = np.zeros(n, np.float32)
    i = int(rng.integers(0, 3000))
    while i < n:
        kind = rng.choice(4, p=[0.35, 0.2, 0.25, 0.2])
        if kind == 0:
            ln = int(rng.integers(300, 6000)); env[i:i + ln] = 1
        elif kind == 1:
            ln = int(rng.integers(900, 9000))
        elif kind == 2:
            ln = int(rng.integers(60, 790))
        else:
            ln = int(rng.integers(8, 45)); env[i:i + ln] = 1
        i += ln
    if boundary_trains:
        for k, b in enumerate(range(UNIT, n, UNIT)):
            a = b - int(rng.integers(2000, 9000))
            env[a - 3000:a] = 0                                          # a long pause opens a fresh group
            j = a
            stop = b + int(rng.integers(500, 6000))
            while j < stop:
                t = int(rng.integers(8, 45)); env[j:j + t] = 1; j += t    # tiny burst: zero symbols
                g = int(rng.integers(60, 700)); env[j:j + g] = 0; j += g  # short pause: zero bits
            if k % 2 == 0:
                env[j:j + 2500] = 1                                       # data after all: the group is a message
                env[j + 2500:j + 5000] = 0
            else:
                env[j:j + 4000] = 0                                       # closed by a long pause: dropped
    iq = synth_fsk(n, sps=100, seed=seed + 1, noise=0.0)
    iq = iq * env[:, None] + 0.012 * rng.standard_normal((n, 2)).astype(np.float32)
    return iq.astype(np.float32)


def _oracle_flat(oracle, iq, p):
    qad = oracle.afp_demod(iq, p.noise_threshold, p.modulation_type, 2 ** p.bits_per_symbol)
    pp = oracle.grab_pulse_lens(qad, p.center, p.tolerance, p.modulation_type, p.samples_per_symbol, p.bits_per_symbol, p.center_spacing)
    return (pp,) + tuple(oracle.ppseq_to_bits_flat(pp, p.samples_per_symbol, p.bits_per_symbol, True, p.pause_threshold))


def _got(r):
    r.check()
    return (r.ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy(), r.bit_sample_pos(), r.pos_offsets())


def _assert_equal(got, ref, tag):
    names = ("rows", "bits", "msg_off", "pauses", "pos", "pos_off")
    for name, g, e in zip(names, got, ref):
        if not np.array_equal(g, e):
            g, e = np.asarray(g), np.asarray(e)
            first = int(np.argmax(g[:min(len(g), len(e))].reshape(min(len(g), len(e)), -1) != e[:min(len(g), len(e))].reshape(min(len(g), len(e)), -1))) \
                if min(len(g), len(e)) else -1
            raise AssertionError(f"{tag}: {name} differs (got {g.shape}, expected {e.shape}, first difference near flat index {first})")


@pytest.fixture()
def one_tile_chunks():
    from urh_amd import _lib
    lib = _lib.load()
    lib.urhgpu_test_force_tiles_per_chunk(1)
    yield
    lib.urhgpu_test_force_tiles_per_chunk(0)


@pytest.mark.parametrize("segments", [2, 3, 5, 8])
@pytest.mark.parametrize("want_pos", [True, False])
def test_segmented_tail_equals_oracle(oracle, one_tile_chunks, segments, want_pos):
    """segments: rows segments (resolve + rows, shipped as they are written) in front of ONE bits segment (tile scan, group scan,
    expansion, pack); several bits segments are what an upload runs (test_upload_piece_by_piece_equals_oracle)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, want_pos)
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": 1, "stream_segments": segments})
    st = pipe.stream(N, p, want_qad=True, want_pos=want_pos)
    caps = [_events_capture(N, 11), _events_capture(N, 12), _events_capture(N, 13, boundary_trains=False),
            synth_fsk(N, sps=100, seed=5, noise=0.04),                                      # one message, no pause at all
            (0.02 * np.random.default_rng(3).standard_normal((N, 2))).astype(np.float32),     # nothing above the noise gate: one pause row
            _events_capture(N, 14)]
    dev = [torch.from_numpy(c).cuda() for c in caps]
    got = {}
    for d in dev:
        r = st.push(d)
        if r is not None:
            got[r.seq] = _got(r)
    for r in st.flush():
        got[r.seq] = _got(r)
    stats = st.stats()
    st.close()
    assert stats["predicted_bytes"] == -len(caps), stats       # every pass took the segmented route
    for i, iq in enumerate(caps):
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i}, {segments} rows segments")


def test_segmented_tail_order4_int16_and_qad(oracle, one_tile_chunks):
    """4-FSK (three thresholds: the order-4 bit planes) on an int16 capture (not on the CU-masked stream), qad of every pass read back"""
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 2, 0.0, 0.0, 0.03, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": 1, "stream_segments": 8})
    caps = [synth_fsk(N, sps=100, seed=40 + i, noise=0.03, pause_every=N // (3 + i), pause_len=3000 + 977 * i, dtype=np.int16) for i in range(4)]
    st = pipe.stream(N, p, want_qad=True, want_pos=True, dtype=np.int16)
    dev = [torch.from_numpy(c).cuda() for c in caps]
    got, qads = {}, {}

    def keep(r):
        got[r.seq] = _got(r)
        q = np.empty(N, np.float32)
        _lib.check(_lib.load().urhgpu_memcpy_to_host(pipe.ctx.handle, C.c_void_p(r.d_qad_ptr), q.ctypes.data_as(C.c_void_p), N * 4))
        qads[r.seq] = q
    for d in dev:
        r = st.push(d)
        if r is not None:
            keep(r)
    for r in st.flush():
        keep(r)
    assert st.stats()["predicted_bytes"] == -len(caps)
    st.close()
    for i, iq in enumerate(caps):
        ref = _oracle_flat(oracle, iq, p)
        _assert_equal(got[i], ref, f"capture {i}")
        qad = oracle.afp_demod(iq, 0.0, "FSK", 4)
        assert np.array_equal(qads[i].view(np.uint32), qad.view(np.uint32)), i


def test_segmented_and_plain_passes_interleave(oracle, one_tile_chunks):
    """captures that qualify (whole tiles) and captures that do not (a partial tile at the end: the ordinary tail, pack kernel and copy)
    through ONE stream: the slots, arenas, progress counters and host blobs of the two routes do not disturb each other"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": 1, "stream_segments": 8})
    st = pipe.stream(N, p, want_qad=False, want_pos=True)
    sizes = [N, N - 777, N, N // 2, N - 2048, N, 70_001, N]
    caps = [_events_capture(N, 60 + i)[:n].copy() for i, n in enumerate(sizes)]
    dev = [torch.from_numpy(c).cuda() for c in caps]
    got = {}
    for d in dev:
        r = st.push(d)
        if r is not None:
            got[r.seq] = _got(r)
    for r in st.flush():
        got[r.seq] = _got(r)
    st.close()
    for i, iq in enumerate(caps):
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i} ({sizes[i]} samples)")


@pytest.mark.parametrize("pieces", [8, 3, 16, 2])
def test_upload_piece_by_piece_equals_oracle(oracle, one_tile_chunks, pieces):
    """urhgpu_stream_push_upload: the capture starts on the HOST (pinned memory); pieces are copied into the device buffer and demodulated
    as they land.  The host receives what the reference computes for the capture, the device buffer holds the capture, the demodulated
    signal of every pass equals the oracle's.  Captures the segmented path does not take (a partial tile at the end) are uploaded in
    one copy in front of an ordinary pass."""
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True, tuning={"upload_pieces": pieces})
    st = pipe.stream(N, p, want_qad=True, want_pos=True)
    sizes = [N, N, N - 777, N, N // 2]
    caps = [_events_capture(N, 70 + i)[:n].copy() for i, n in enumerate(sizes)]
    host = [torch.from_numpy(c).pin_memory() for c in caps]
    dev = [torch.zeros_like(h, device="cuda") for h in host]
    got, qads = {}, {}

    def keep(r):
        got[r.seq] = _got(r)
        n = sizes[r.seq]
        q = np.empty(n, np.float32)
        _lib.check(_lib.load().urhgpu_memcpy_to_host(pipe.ctx.handle, C.c_void_p(r.d_qad_ptr), q.ctypes.data_as(C.c_void_p), n * 4))
        qads[r.seq] = q
    for h, d in zip(host, dev):
        r = st.push_upload(h, d)
        for x in ([r] if r is not None else []) + st.flush():       # one capture at a time: every result's qad is read before the next pass
            keep(x)
    st.close()
    torch.cuda.synchronize()
    for i, iq in enumerate(caps):
        assert np.array_equal(dev[i].cpu().numpy().view(np.uint32), iq.view(np.uint32)), f"capture {i}: the device buffer does not hold the capture"
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i} ({sizes[i]} samples, {pieces} pieces)")
        qad = oracle.afp_demod(iq, p.noise_threshold, "FSK", 2)
        assert np.array_equal(qads[i].view(np.uint32), qad.view(np.uint32)), f"capture {i}: qad"


def test_upload_back_to_back(oracle, one_tile_chunks):
    """uploads pushed back to back (results handed out three pushes later), numpy sources (pageable memory: slower, same results)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, False)
    pipe = DevicePipeline(0, pipelined=True)
    st = pipe.stream(N, p, want_qad=False, want_pos=False)
    caps = [_events_capture(N, 80 + i) for i in range(5)]
    dev = [torch.zeros(N, 2, dtype=torch.float32, device="cuda") for _ in caps]
    got = {}
    for c, d in zip(caps, dev):
        r = st.push_upload(c, d)
        if r is not None:
            got[r.seq] = _got(r)
    for r in st.flush():
        got[r.seq] = _got(r)
    st.close()
    for i, iq in enumerate(caps):
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i}")


@pytest.mark.parametrize("policy,want_pos,pos_direct", [(3, True, 1), (3, False, 1), (4, True, 1), (3, True, 0), (5, True, 1), (5, True, 0)])
def test_direct_passes_equal_oracle(oracle, one_tile_chunks, policy, want_pos, pos_direct):
    """stream_policy 3 / 4: passes whose tail is ONE segment behind the hot kernel (an event, no gate) and stores rows and packed
    results into the pinned host blob itself -- back to back, whole-tile and partial-tile captures mixed (the latter: pack + copy)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, want_pos)
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": policy, "stream_pos_direct": pos_direct})
    st = pipe.stream(N, p, want_qad=True, want_pos=want_pos)
    sizes = [N, N, N - 2048, N, N - 777, N, N // 2, N]
    caps = [_events_capture(N, 120 + i)[:n].copy() for i, n in enumerate(sizes)]
    dev = [torch.from_numpy(c).cuda() for c in caps]
    got = {}
    for d in dev:
        r = st.push(d)
        if r is not None:
            got[r.seq] = _got(r)
    for r in st.flush():
        got[r.seq] = _got(r)
    stats = st.stats()
    st.close()
    if not (policy == 5 and not pos_direct):               # (policy 5 without direct positions: back-to-back passes go through pack + copy engine)
        assert -stats["predicted_bytes"] >= sum(1 for n in sizes if n % 2048 == 0) - 1, stats      # the whole-tile captures took the kernel-written route
    for i, iq in enumerate(caps):
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i} ({sizes[i]} samples), policy {policy}")


def test_upload_and_resident_pushes_interleave_order4_and_tiny(oracle, one_tile_chunks):
    """uploads and resident pushes through ONE stream, back to back without a flush in between; 4-FSK (the order-4 bit planes); captures
    of a few samples (shorter than a tile: one copy + the state-byte kernel) -- every result equals the oracle, every device buffer holds
    its capture"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 2, 0.05, 0.0, 0.03, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True)
    st = pipe.stream(N, p, want_qad=False, want_pos=True)
    sizes = [N, 5, N, N // 2, 2047, N, 3, N - 4096]
    caps = [synth_fsk(N, sps=100, seed=140 + i, noise=0.03, pause_every=N // (3 + i % 3), pause_len=2500 + 311 * i)[:n].copy() for i, n in enumerate(sizes)]
    host = [torch.from_numpy(c).pin_memory() for c in caps]
    dev = [torch.zeros_like(h, device="cuda") for h in host]
    got = {}
    for i, (h, d) in enumerate(zip(host, dev)):
        if i % 3 == 2:                                    # every third capture is resident already
            d.copy_(h)
            r = st.push(d)
        else:
            r = st.push_upload(h, d)
        if r is not None:
            got[r.seq] = _got(r)
    for r in st.flush():
        got[r.seq] = _got(r)
    st.close()
    torch.cuda.synchronize()
    for i, iq in enumerate(caps):
        assert np.array_equal(dev[i].cpu().numpy().view(np.uint32), iq.view(np.uint32)), i
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i} ({sizes[i]} samples)")


@pytest.mark.parametrize("policy", [5, 1])
def test_stream_randomised_vs_oracle(oracle, one_tile_chunks, policy):
    """differential fuzz of the stream routes: random parameter sets (tolerance, samples per symbol, noise gate, center, pause threshold,
    orders 2 and 4, float32 / int16 / int8 captures), a stream each, captures of random length pushed back to back -- whole tiles (direct
    passes / segments where the capture is long enough), partial tiles and captures shorter than a tile (pack + copy)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    rng = np.random.default_rng(4242 + policy)
    pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": policy, "stream_segments": 4})
    n_max = 1 << 21
    for s_i in range(10):
        dtype = [np.float32, np.int16, np.float32, np.int8][s_i % 4]
        sps = int(rng.choice([5, 17, 100, 333]))
        tol = int(rng.choice([0, 1, 5, 9, 40]))
        bps, spacing = ((2, float(rng.choice([0.05, 0.3]))) if s_i % 3 == 2 else (1, 1.0))
        want_pos = bool(s_i % 2)
        scale = 1.0 if dtype == np.float32 else float(np.iinfo(dtype).max) * 0.7
        p = DemodParams("FSK", bps, float(rng.choice([0.0, 0.2])) * scale, float(rng.choice([0.0, 0.1, -0.2])), spacing, tol, sps, 0.1,
                        int(rng.choice([0, 1, 8])), want_pos)
        st = pipe.stream(n_max, p, want_qad=False, want_pos=want_pos, dtype=dtype, cap_rows=n_max // (tol + 1) + 2)     # (worst case: noise only)
        sizes = [int(x) for x in rng.choice([2048 * int(rng.integers(1, 1024)), 2048 * int(rng.integers(512, 1024)), int(rng.integers(3, 5000)),
                                             int(rng.integers(5000, 900_000)), n_max], size=7)]
        caps = [synth_fsk(n, sps=sps, seed=1000 * s_i + k, noise=float(rng.choice([0.0, 0.03, 0.3])), pause_every=int(rng.choice([0, max(n // 3, 1), 2500])),
                          pause_len=int(rng.choice([7, 130, 2100])), dtype=dtype) for k, n in enumerate(sizes)]
        dev = [torch.from_numpy(c).cuda() for c in caps]
        got = {}
        for d in dev:
            r = st.push(d)
            if r is not None:
                got[r.seq] = _got(r) if want_pos else (r.check().ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy())
        for r in st.flush():
            got[r.seq] = _got(r) if want_pos else (r.check().ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy())
        st.close()
        for k, iq in enumerate(caps):
            ref = _oracle_flat(oracle, iq, p)
            _assert_equal(got[k], ref[:len(got[k])], f"stream {s_i} ({np.dtype(dtype).name}, sps {sps}, tol {tol}, order {2 ** bps}) capture {k} ({sizes[k]} samples)")


def test_latency_setting_one_capture_at_a_time(oracle, one_tile_chunks):
    """CaptureStream(latency=True): every capture pushed and flushed on its own finds the pipeline idle and runs its tail in the default
    segmentation (7 rows segments, 1 bits segment); two pushed back to back: the second one direct.  All equal to the oracle."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
    pipe = DevicePipeline(0, pipelined=True)
    st = pipe.stream(N, p, want_qad=False, want_pos=True, latency=True)
    caps = [_events_capture(N, 160 + i) for i in range(4)]
    dev = [torch.from_numpy(c).cuda() for c in caps]
    got = {}
    for d in dev[:2]:
        assert st.push(d) is None
        (r,) = st.flush()
        got[r.seq] = _got(r)
    for d in dev[2:]:
        st.push(d)
    for r in st.flush():
        got[r.seq] = _got(r)
    st.close()
    pipe.ctx.set_tuning("stream_latency", 0)
    for i, iq in enumerate(caps):
        _assert_equal(got[i], _oracle_flat(oracle, iq, p), f"capture {i}")


def _synth_ask(n, sps, seed, noise, pause_every, pause_len, dtype):
    """seeded OOK-like capture: a carrier keyed by random symbols (amplitudes 1.0 / 0.15), AWGN, optional silent gaps"""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, n // sps + 1)
    env = np.repeat(np.where(bits == 1, 1.0, 0.15), sps)[:n]
    phase = 2 * np.pi * 0.013 * np.arange(n)
    iq = np.stack([env * np.cos(phase), env * np.sin(phase)], axis=1)
    if pause_every:
        for a in range(pause_every, n, pause_every + pause_len):
            iq[a:a + pause_len] = 0
    iq = iq + noise * rng.standard_normal((n, 2))
    if np.dtype(dtype) == np.float32:
        return iq.astype(np.float32)
    info = np.iinfo(dtype)
    scale = (info.max - info.min) / 2 * 0.7
    off = (info.max + info.min + 1) / 2
    return np.clip(np.round(iq * scale + off), info.min, info.max).astype(dtype)


def test_stream_fuzz_extended(oracle, one_tile_chunks):
    """the long form of the differential fuzz (URH_FUZZ_ROUNDS rounds, default 6; URH_FUZZ_SEED): FSK and ASK, all five sample types, every
    stream policy and segment count, the latency setting, qad wanted or not, positions shipped or not -- a stream per round, seven captures
    of random length each, all against the oracle (signal_functions.pyx:333-495, ProtocolAnalyzer.py:262-330)"""
    import os
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    rounds = int(os.environ.get("URH_FUZZ_ROUNDS", "6"))
    seed0 = int(os.environ.get("URH_FUZZ_SEED", "0"))
    n_max = 1 << 21
    for s_i in range(rounds):
        rng = np.random.default_rng([777, seed0, s_i])
        policy = int(rng.choice([5, 1, 3, 4, 0, 2]))
        segs = int(rng.integers(1, 10))
        pipe = DevicePipeline(0, pipelined=True, tuning={"stream_policy": policy, "stream_segments": segs})
        dtype = [np.float32, np.int16, np.int8, np.uint8, np.uint16][int(rng.integers(0, 5))]
        mod = "FSK" if rng.random() < 0.6 else "ASK"
        sps = int(rng.choice([5, 17, 100, 333]))
        tol = int(rng.choice([0, 1, 5, 9, 40]))
        bps, spacing = ((2, float(rng.choice([0.05, 0.3]))) if (mod == "FSK" and rng.random() < 0.3) else (1, 1.0))
        want_pos = bool(rng.integers(0, 2))
        want_qad = bool(rng.integers(0, 2))
        latency = bool(rng.integers(0, 2))
        scale = 1.0 if dtype == np.float32 else (float(np.iinfo(dtype).max) - float(np.iinfo(dtype).min)) / 2 * 0.7
        if mod == "FSK":
            noise_thr, center = float(rng.choice([0.0, 0.2])) * scale, float(rng.choice([0.0, 0.1, -0.2]))
        else:
            noise_thr, center = float(rng.choice([0.0, 0.05])) * scale, float(rng.choice([0.5, 0.4, 0.7])) * scale
        p = DemodParams(mod, bps, noise_thr, center, spacing, tol, sps, 0.1, int(rng.choice([0, 1, 8])), want_pos)
        st = pipe.stream(n_max, p, want_qad=want_qad, want_pos=want_pos, dtype=dtype, cap_rows=n_max // (tol + 1) + 2, latency=latency)
        sizes = [int(x) for x in rng.choice([2048 * int(rng.integers(1, 1024)), 2048 * int(rng.integers(512, 1024)), int(rng.integers(3, 5000)),
                                             int(rng.integers(5000, 900_000)), n_max], size=7)]
        gen = synth_fsk if mod == "FSK" else None
        caps = []
        for k, n in enumerate(sizes):
            nz = float(rng.choice([0.0, 0.03, 0.3])); pe = int(rng.choice([0, max(n // 3, 1), 2500])); pl = int(rng.choice([7, 130, 2100]))
            caps.append(synth_fsk(n, sps=sps, seed=100000 * seed0 + 1000 * s_i + k, noise=nz, pause_every=pe, pause_len=pl, dtype=dtype) if gen
                        else _synth_ask(n, sps, 100000 * seed0 + 1000 * s_i + k, nz, pe, pl, dtype))
        dev = [torch.from_numpy(c).cuda() for c in caps]
        take = (lambda r: _got(r)) if want_pos else (lambda r: (r.check().ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy()))
        got = {}
        for j, d in enumerate(dev):
            r = st.push(d)
            if r is not None:
                got[r.seq] = take(r)
            if latency and rng.random() < 0.5:
                for r in st.flush():
                    got[r.seq] = take(r)
        for r in st.flush():
            got[r.seq] = take(r)
        st.close()
        tag0 = (f"round {s_i} seed {seed0} ({mod}, {np.dtype(dtype).name}, sps {sps}, tol {tol}, order {2 ** bps}, policy {policy}, segments {segs}, "
                f"latency {latency}, qad {want_qad}, pos {want_pos})")
        for k, iq in enumerate(caps):
            ref = _oracle_flat(oracle, iq, p)
            _assert_equal(got[k], ref[:len(got[k])], f"{tag0} capture {k} ({sizes[k]} samples)")
        del pipe
