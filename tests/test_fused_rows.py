"""The row stage INSIDE the hot kernel (launchers.hpp: FusedRows; demod_runs.hip, FUSED instantiation): every chunk's wavefront 0 publishes
what its chunk does to the reference's state machine, looks back over its predecessors and writes the chunk's pulse-table rows, the
host blob's row sections and its tile's bit aggregates itself -- a chunk owns the pending run of the chunk BEFORE it.  Whatever the
chunk plan, the tolerance, the order, the sample type or the capture (runs that span many chunks, captures without a single state
change, rows of thousands of bits), the result equals the oracle's (= the reference's: tests/test_oracle.py) and the un-fused tail's."""
import numpy as np
import pytest

from conftest import synth_fsk
from test_stream_segments import _assert_equal, _events_capture, _got, _oracle_flat

pytestmark = pytest.mark.gpu

N = 1 << 21


@pytest.fixture()
def tiles(request):
    from urh_amd import _lib
    lib = _lib.load()
    lib.urhgpu_test_force_tiles_per_chunk(request.param)
    yield request.param
    lib.urhgpu_test_force_tiles_per_chunk(0)


def _captures(n, dtype=np.float32):
    rng = np.random.default_rng(77)
    caps = {
        "bursts": _events_capture(n, 501),
        "plain": synth_fsk(n, sps=100, seed=3, noise=0.05),
        "long pauses": synth_fsk(n, sps=100, seed=4, noise=0.03, pause_every=n // 7, pause_len=70_000),
        "noise only": (0.02 * rng.standard_normal((n, 2))).astype(np.float32),
        "one run": np.stack([np.cos(0.13 * np.arange(n)), np.sin(0.13 * np.arange(n))], axis=1).astype(np.float32),
        "short symbols": synth_fsk(n, sps=7, seed=5, noise=0.08),
    }
    if np.dtype(dtype) != np.float32:
        info = np.iinfo(dtype)
        caps = {k: np.clip(np.round(v * (info.max * 0.6)), info.min, info.max).astype(dtype) for k, v in caps.items()}
    return caps


def _device_result(pipe, iq, p, want_qad=True):
    import torch
    res = pipe.iq_to_bits_checked(torch.from_numpy(iq).cuda(), p, want_qad=want_qad)
    return (res.ppseq(),) + tuple(res.flat()), (res.qad.cpu().numpy() if want_qad else None)


@pytest.mark.parametrize("tiles", [1, 2, 4], indirect=True)
@pytest.mark.parametrize("tol", [0, 1, 5, 33, 64])
def test_fused_rows_equal_oracle_and_unfused(oracle, tiles, tol):
    from urh_amd.pipeline import DemodParams, DevicePipeline
    fused, plain = DevicePipeline(0), DevicePipeline(0, tuning={"hot_fused_rows": 0})
    for name, iq in _captures(N).items():
        sps = 7 if name == "short symbols" else 100
        p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, tol, sps, 0.1, 8, True)
        got, qad = _device_result(fused, iq, p)
        want = _oracle_flat(oracle, iq, p)
        _assert_equal(got, want, f"{name}, tolerance {tol}, {tiles} tiles per chunk")
        ref_qad = oracle.afp_demod(iq, p.noise_threshold, "FSK", 2)
        assert np.array_equal(qad.view(np.uint32), ref_qad.view(np.uint32)), name
        got2, _ = _device_result(plain, iq, p, want_qad=False)
        _assert_equal(got2, want, f"un-fused: {name}, tolerance {tol}")


@pytest.mark.parametrize("tiles", [1, 4], indirect=True)
@pytest.mark.parametrize("dtype,bps", [(np.int16, 1), (np.int8, 1), (np.float32, 2), (np.int16, 2), (np.uint8, 1)])
def test_fused_rows_orders_and_sample_types(oracle, tiles, dtype, bps):
    from urh_amd.pipeline import DemodParams, DevicePipeline
    pipe = DevicePipeline(0)
    for seed in range(3):
        iq = synth_fsk(N, sps=100, seed=20 + seed, noise=0.04, pause_every=N // (4 + seed), pause_len=9000 + 4000 * seed, dtype=dtype)
        scale = 1.0 if dtype == np.float32 else float(np.abs(iq.astype(np.float64) - (128 if dtype == np.uint8 else 0)).max())
        p = DemodParams("FSK", bps, 0.0 if dtype == np.uint8 else 0.1 * scale, 0.0, 0.03 if bps == 2 else 1.0, 5, 100, 0.1, 8, True)
        got, qad = _device_result(pipe, iq, p)
        _assert_equal(got, _oracle_flat(oracle, iq, p), f"{np.dtype(dtype).name}, {bps} bits per symbol, seed {seed}")
        assert np.array_equal(qad.view(np.uint32), oracle.afp_demod(iq, p.noise_threshold, "FSK", 2 ** bps).view(np.uint32))


@pytest.mark.parametrize("tiles", [1], indirect=True)
def test_fused_rows_sizes_and_huge_rows(oracle, tiles):
    """captures of one chunk, two chunks, an odd number of tiles; rows of more than 4096 bits (one sample per symbol: a run is its length in
    bits), which the hot kernel lists for the expansion's extra workgroups"""
    from urh_amd.pipeline import DemodParams, DevicePipeline
    pipe = DevicePipeline(0)
    for n in (2048, 4096, 2048 * 7, 2048 * 257):
        iq = synth_fsk(n, sps=100, seed=n % 97, noise=0.04, pause_every=max(n // 3, 1), pause_len=min(3000, n // 4))
        p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
        got, _ = _device_result(pipe, iq, p)
        _assert_equal(got, _oracle_flat(oracle, iq, p), f"{n} samples")
    iq = synth_fsk(1 << 19, sps=20_000, seed=8, noise=0.02)                  # runs of 20 000 samples
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 1, 0.1, 50_000, True)        # one sample per symbol, nothing is a long pause
    got, _ = _device_result(pipe, iq, p)
    want = _oracle_flat(oracle, iq, p)
    assert want[0][:, 1].max() > 4096
    _assert_equal(got, want, "rows of thousands of bits")


@pytest.mark.parametrize("tiles", [1, 4], indirect=True)
@pytest.mark.parametrize("want_pos", [False, True])
def test_fused_rows_through_the_capture_stream(oracle, tiles, want_pos):
    """the same through urhgpu_stream_*: rows stored into the pinned host blob by the hot kernel itself, pass after pass (three scratch arenas
    and descriptor tags in rotation), whole-tile captures mixed with ones that end in a partial tile (un-fused: pack + copy)"""
    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, want_pos)
    pipe = DevicePipeline(0, pipelined=True)
    st = pipe.stream(N, p, want_qad=True, want_pos=want_pos)
    sizes = [N, N, N - 2048, N - 777, N, N // 2, N, N]
    caps = [_events_capture(N, 300 + i)[:n].copy() for i, n in enumerate(sizes)]
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
