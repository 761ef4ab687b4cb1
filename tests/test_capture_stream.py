"""urhgpu_stream_* (capture after capture, compact results on the host; include/urhgpu.h) against the oracle: every pass of a stream of
DIFFERENT captures -- sizes, pauses, modulations, bits per symbol -- comes back with the pulse table, bits, pauses, offsets and
bit_sample_pos the reference computes for that capture, whatever was in flight around it (hot kernel of the next pass, tail of this
one, copy of the previous one)."""
import numpy as np
import pytest

from conftest import synth_fsk

pytestmark = pytest.mark.gpu


def _captures(mod, bps, k, n_max, rng):
    out = []
    for i in range(k):
        n = int(rng.choice([n_max, n_max - 777, n_max // 2 + 13, 4096 * 3, 70_001]))
        iq = synth_fsk(n, sps=100, seed=100 + i, noise=0.04, pause_every=max(n // (2 + i % 3), 5000), pause_len=n // 19 + 901)
        if mod == "ASK":
            env = np.repeat(np.random.default_rng(i).integers(0, 2 ** bps, n // 100 + 1), 100)[:n] / (2 ** bps - 1)
            iq = (iq * (0.05 + 0.95 * env)[:, None]).astype(np.float32)
        out.append(iq)
    return out


@pytest.mark.parametrize("mod,bps,want_pos", [("FSK", 1, True), ("FSK", 1, False), ("ASK", 1, True), ("FSK", 2, True)])
def test_stream_of_different_captures_equals_oracle(oracle, mod, bps, want_pos):
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    rng = np.random.default_rng(7)
    n_max = (1 << 21) + 4096
    pipe = DevicePipeline(0, pipelined=True)
    center = 0.0 if mod == "FSK" else 0.4
    spacing = 1.0 if bps == 1 else 0.03
    p = DemodParams(mod, bps, 0.1, center, spacing, 5, 100, 0.1, 8, want_pos)
    caps = _captures(mod, bps, 9, n_max, rng)
    dev = [torch.from_numpy(c).cuda() for c in caps]
    st = pipe.stream(n_max, p, want_qad=True, want_pos=want_pos)
    got = {}

    def keep(r):
        if r is not None:
            r.check()
            got[r.seq] = (r.ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy(), r.bit_sample_pos(), r.pos_offsets(), r.blob_bytes, r.n_samples)
    for d in dev:
        keep(st.push(d))
    for r in st.flush():
        keep(r)
    assert sorted(got) == list(range(len(caps)))
    for i, iq in enumerate(caps):
        qad = oracle.afp_demod(iq, 0.1, mod, 2 ** bps)
        pp = oracle.grab_pulse_lens(qad, center, 5, mod, 100, bps, spacing)
        bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(pp, 100, bps, True, 8)
        g = got[i]
        assert g[7] == len(iq)
        assert np.array_equal(g[0], pp), i
        assert np.array_equal(g[1], bits) and np.array_equal(g[2], off) and np.array_equal(g[3], pauses), i
        # positions: the device's (want_pos) or derived on the host from the shipped pulse table (HostBits.bit_sample_pos): the same
        assert np.array_equal(g[4], pos) and np.array_equal(g[5], poff), i
        # what crossed PCIe: at most 5 B per row (3 + escapes with 16-bit lengths), 1 bit per bit, 4 B per position, 24 B per message (+ header and alignment)
        assert g[6] <= 5 * len(pp) + len(bits) // 8 + 4 * len(pos) * (1 if want_pos else 0) + 24 * len(pauses) + 512, (i, g[6])
    st.close()
    # the pipeline is usable as before afterwards
    res = pipe.iq_to_bits(dev[0], p, want_qad=True)
    assert np.array_equal(res.ppseq(), got[0][0])


def test_stream_reports_a_truncated_pulse_table(oracle):
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = 1 << 20
    iq = (0.3 * np.random.default_rng(1).standard_normal((n, 2))).astype(np.float32)        # noise only: a row every few samples
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 1, 100, 0.1, 8, True)
    pipe = DevicePipeline(0)
    st = pipe.stream(n, p)
    d = torch.from_numpy(iq).cuda()
    assert st.push(d) is None
    (r,) = st.flush()
    pp = oracle.grab_pulse_lens(oracle.afp_demod(iq, 0.0, "FSK", 2), 0.0, 1, "FSK", 100, 1, 1.0)
    assert r.truncated and r.rows_needed == len(pp)
    with pytest.raises(Exception):
        r.check()
    st.close()
    st = pipe.stream(n, p, cap_rows=r.rows_needed + 8)
    st.push(d)
    (r,) = st.flush()
    assert not r.truncated and np.array_equal(r.ppseq(), pp)
    st.close()


def test_stream_pushed_from_a_side_stream_with_the_capture_produced_there(oracle):
    """The hot kernel of a pipelined pass runs on a private CU-masked stream of the context: work the caller has queued on ITS stream --
    here a kernel that produces the capture on a non-default torch stream right before every push -- is ordered before it through an
    event (from torch's default = NULL stream the runtime's own NULL-stream ordering does it)."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = 1 << 21
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, False)
    caps = [synth_fsk(n, sps=100, seed=300 + i, noise=0.04, pause_every=n // 3, pause_len=n // 23) for i in range(6)]
    side = torch.cuda.Stream()
    got = {}
    with torch.cuda.stream(side):
        pipe = DevicePipeline(0, pipelined=True)
        st = pipe.stream(n, p, want_qad=False, want_pos=False)
        staging = [torch.from_numpy(c).cuda() for c in caps]
        work = [torch.empty_like(staging[0]) for _ in range(len(caps))]      # (an input buffer must stay untouched until its pass has run)
        torch.cuda.synchronize()
        for i in range(len(caps)):
            buf = work[i]
            buf.copy_(staging[i])                       # a kernel on the side stream: the capture is not there before it has run
            r = st.push(buf)
            if r is not None:
                got[r.seq] = (r.check().ppseq(), r.bits(), r.pauses.copy())
        for r in st.flush():
            got[r.seq] = (r.check().ppseq(), r.bits(), r.pauses.copy())
        st.close()
    for i, iq in enumerate(caps):
        pp = oracle.grab_pulse_lens(oracle.afp_demod(iq, 0.1, "FSK", 2), 0.0, 5, "FSK", 100, 1, 1.0)
        bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
        assert np.array_equal(got[i][0], pp) and np.array_equal(got[i][1], bits) and np.array_equal(got[i][2], pauses), i


def test_stream_ships_16_bit_lengths_with_escapes(oracle):
    """Staged passes (round 6) ship 3 bytes per pulse-table row -- uint16 lengths, rows of 65535 samples and more through the escape list
    (include/urhgpu.h: URHGPU_BLOB_LEN16): captures with pauses of 70 000 .. 300 000 samples (one of exactly 65 535 and one of 65 534: the boundary),
    pushed back to back, come back with the reference's pulse table; the blob says how many rows took the list."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = (1 << 20) + 4096 * 5
    pipe = DevicePipeline(0, pipelined=True)
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, False)
    caps = []
    for i, gaps in enumerate(([(50_000, 70_000)], [(10_000, 65_535 + 11), (300_000, 65_534 + 11), (500_000, 300_000)], [], [(1000, 200_000), (400_000, 66_000)])):
        iq = synth_fsk(n, sps=100, seed=300 + i, noise=0.02)
        for a, ln in gaps:
            iq[a:a + ln] = 0.0                                         # below the noise gate: one PAUSE row of about ln samples
        caps.append(iq)
    st = pipe.stream(n, p, want_qad=True, want_pos=False)
    got = {}

    def keep(r):
        if r is not None:
            r.check()
            got[r.seq] = (r.ppseq(), r.bits(), r.pauses.copy(), r._len16[2] if r._len16 is not None else -1, r.blob_bytes)
    for rep in range(2):
        for c in caps:
            keep(st.push(torch.from_numpy(c).cuda()))
    for r in st.flush():
        keep(r)
    assert sorted(got) == list(range(2 * len(caps)))
    for k in range(2 * len(caps)):
        iq = caps[k % len(caps)]
        pp = oracle.grab_pulse_lens(oracle.afp_demod(iq, 0.1, "FSK", 2), 0.0, 5, "FSK", 100, 1, 1.0)
        bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
        g = got[k]
        assert np.array_equal(g[0], pp) and np.array_equal(g[1], bits) and np.array_equal(g[2], pauses), k
        assert g[3] == int((pp[:, 1] >= 0xFFFF).sum() + (pp[:, 1] < 0).sum()), (k, g[3])          # 16-bit lengths were shipped, the long rows escaped
        assert g[4] <= 3 * len(pp) + 8 * g[3] + len(bits) // 8 + 24 * len(pauses) + 512, (k, g[4])
    assert any(got[k][3] > 0 for k in got)
    st.close()


@pytest.mark.parametrize("bps", [1, 2])
def test_stream_ships_packed_rows_for_dense_tables(oracle, bps):
    """A stream whose last result held more than one pulse-table row per 64 samples ships state and length of a row in ONE uint16 (include/urhgpu.h:
    URHGPU_BLOB_ROW16: 2 bytes per row; orders 2 and 4): captures at 8 samples per symbol with pauses of 8191 samples and more (one of exactly
    8191 and one of 8190: the boundary of the escape list), a sparse capture in between (the stream goes back to 16-bit lengths for the pass
    after it), pushed back to back, come back with the reference's pulse table, bits and pauses."""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = (1 << 20) + 4096 * 3
    pipe = DevicePipeline(0, pipelined=True)
    p = DemodParams("FSK", bps, 0.1, 0.0, 0.05 if bps == 2 else 1.0, 1, 8, 0.1, 8, False)
    caps = []
    for i, (sps, gaps) in enumerate(((8, [(50_000, 9000)]), (8, [(10_000, 8191 + 3), (300_000, 8190 + 3), (500_000, 300_000)]), (400, []),
                                     (8, []), (8, [(1000, 70_000), (400_000, 8300)]))):
        iq = synth_fsk(n, sps=sps, seed=500 + i, noise=0.02, deviation_hz=30e3)
        for a, ln in gaps:
            iq[a:a + ln] = 0.0                                         # below the noise gate: one PAUSE row of about ln samples
        caps.append(iq)
    st = pipe.stream(n, p, want_qad=True, want_pos=False)
    got = {}

    def keep(r):
        if r is not None:
            r.check()
            got[r.seq] = (r.ppseq(), r.bits(), r.pauses.copy(), r._row16 is not None, r._row16[2] if r._row16 is not None else -1, r.blob_bytes)
    for rep in range(2):
        for c in caps:
            keep(st.push(torch.from_numpy(c).cuda()))
    for r in st.flush():
        keep(r)
    assert sorted(got) == list(range(2 * len(caps)))
    packed = 0
    for k in range(2 * len(caps)):
        iq = caps[k % len(caps)]
        pp = oracle.grab_pulse_lens(oracle.afp_demod(iq, 0.1, "FSK", 1 << bps), 0.0, 1, "FSK", 8, bps, p.center_spacing)
        bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(pp, 8, bps, True, 8)
        g = got[k]
        assert np.array_equal(g[0], pp) and np.array_equal(g[1], bits) and np.array_equal(g[2], pauses), k
        if g[3]:
            packed += 1
            assert g[4] == int((pp[:, 1] >= 0x1FFF).sum() + (pp[:, 1] < 0).sum()), (k, g[4])      # the long rows took the list
            assert g[5] <= 2 * len(pp) + 8 * g[4] + len(bits) // 8 + 24 * len(pauses) + 512, (k, g[5])
    # the pass after a dense result is packed (the stream decides by the last result it has handed out: a few passes behind)
    assert packed >= 3 and any(got[k][3] and got[k][4] > 0 for k in got) and not all(got[k][3] for k in got)
    st.close()


@pytest.mark.parametrize("dtype", [np.int8, np.int16])
def test_integer_stream_switches_to_the_wide_instantiation_and_back(oracle, dtype):
    """Streams of signed integer FSK captures probe their captures (k_wide_probe behind every pass: the share of phase steps beyond the fast loop's
    window) and take the hot kernel's instantiation with the wide loop from 1 % on, the default one again below 0.3 %: narrow captures, then
    wide ones (+-120 kHz at 1 MS/s: 0.75 rad per sample), then narrow ones again, pushed back to back -- every result equals the reference's
    demodulated signal, pulse table, bits and pauses whichever instantiation its pass took."""
    import ctypes as C
    import torch
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    n = (1 << 20) + 4096 * 2
    pipe = DevicePipeline(0, pipelined=True)
    amp = 0.6 * np.iinfo(dtype).max
    p = DemodParams("FSK", 1, 0.1 * amp, 0.0, 1.0, 3, 50, 0.1, 8, False)
    caps = []
    for i, dev_hz in enumerate([20e3] * 4 + [120e3] * 6 + [20e3] * 5):
        x = synth_fsk(n, sps=50, seed=900 + i, noise=0.03, pause_every=n // 3, pause_len=2500, deviation_hz=dev_hz)
        caps.append(np.clip(np.round(x * amp), np.iinfo(dtype).min, np.iinfo(dtype).max).astype(dtype))
    st = pipe.stream(n, p, want_qad=True, want_pos=False, dtype=dtype)
    got = {}

    def keep(r):
        if r is not None:
            r.check()
            q = np.empty(n, np.float32)
            _lib.check(_lib.load().urhgpu_memcpy_to_host(pipe.ctx.handle, C.c_void_p(r.d_qad_ptr), q.ctypes.data_as(C.c_void_p), n * 4))
            got[r.seq] = (q, r.ppseq(), r.bits(), r.pauses.copy())
    wide_after = []
    for c in caps:
        keep(st.push(torch.from_numpy(c).cuda()))
        wide_after.append(st.stats()["wide_passes"])
    for r in st.flush():
        keep(r)
    assert sorted(got) == list(range(len(caps)))
    # the narrow captures at the start took the default instantiation, some of the wide ones (the probe reports a pass late) the wide one,
    # and the stream went back: nothing more is counted at the end
    assert wide_after[3] == 0 and 1 <= wide_after[-1] <= 8 and wide_after[-1] == wide_after[-3], wide_after
    for k, iq in enumerate(caps):
        qad = oracle.afp_demod(iq, p.noise_threshold, "FSK", 2)
        pp = oracle.grab_pulse_lens(qad, 0.0, 3, "FSK", 50, 1, 1.0)
        bits, off, pauses, pos, poff = oracle.ppseq_to_bits_flat(pp, 50, 1, True, 8)
        g = got[k]
        assert np.array_equal(g[0].view(np.uint32), qad.view(np.uint32)), (k, int((g[0].view(np.uint32) != qad.view(np.uint32)).sum()))
        assert np.array_equal(g[1], pp) and np.array_equal(g[2], bits) and np.array_equal(g[3], pauses), k
    st.close()
