"""The sharded IQ->bits protocol (urh_amd/sharding.py): halo / shard-summary / merge / message-flag
all-gathers around per-rank engines.  On the CPU the engines are the executable model of the HIP kernels
(tests/model_shard.py); the stitched result must equal the oracle's single-pass result bit for bit.
  * W ranks as threads (ThreadComm) over many randomised captures and shard boundaries,
  * the real multi-process path: world_size-2 torch.distributed group on gloo.
The GPU version of the same check (HIP engines, W simulated ranks on one MI355X) is in test_gpu_parity.py."""
import os
import threading

import numpy as np
import pytest

import model_shard
import torch
from urh_amd.pipeline import DemodParams
from urh_amd.sharding import ShardedPipeline, ThreadComm, TorchDistComm, shard_bounds, stitch


def _signal(rng, n, mod, bps, noise_val):
    levels = rng.choice([-1.0, -0.3, 0.3, 1.0] if bps == 2 else [-0.5, 0.5], size=n // max(1, int(rng.integers(1, 30))) + 1)
    x = np.repeat(levels, n // len(levels) + 1)[:n].astype(np.float32)
    x[rng.random(n) < rng.choice([0, 0.02, 0.1, 0.3])] *= -1
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n))
        x[a:min(n, a + int(rng.integers(1, 120)))] = noise_val
    if mod == "ASK":
        x = np.abs(x)
    return x


def run_threads(world, make_engine, shards, bounds, n_total, p, halos=None):
    """Run the sharded protocol with `world` ranks as threads; returns the per-rank results.
    halos: per rank, the samples before its shard (None for rank 0) -> no halo exchange (halo_given)."""
    shared = ThreadComm.Shared(world)
    out, err = [None] * world, []

    def work(r):
        try:
            pipe = ShardedPipeline(make_engine(r), ThreadComm(shared, r))
            out[r] = pipe.iq_to_bits(shards[r], p, want_qad=True, pos_base=bounds[r][0], n_total=n_total,
                                     halo_given=halos is not None, left_halo=halos[r] if halos is not None else None)
        except BaseException as e:          # noqa: BLE001 -- re-raised in the main thread
            err.append(e)
            shared.barrier.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if err:
        raise err[0]
    return out


def reference_result(oracle, x, p):
    pp = oracle.grab_pulse_lens(x, p.center, p.tolerance, p.modulation_type, p.samples_per_symbol, p.bits_per_symbol,
                                p.center_spacing)
    return (pp,) + tuple(oracle.ppseq_to_bits_flat(pp, p.samples_per_symbol, p.bits_per_symbol, True, p.pause_threshold))


def assert_same(got, want, tag):
    names = ("ppseq", "bits", "msg_off", "pauses", "pos", "pos_off")
    for k, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), (tag, names[k], a[:12].tolist(), b[:12].tolist(), len(a), len(b))


def test_sharded_model_equals_oracle(oracle):
    rng = np.random.default_rng(17)
    for it in range(int(os.environ.get("SHARD_ITERS", "400"))):
        world = int(rng.choice([2, 3, 4, 8]))
        n = int(rng.integers(2 * world, 900))
        mod = ["ASK", "FSK", "PSK"][it % 3]
        bps = int(rng.integers(1, 3))
        tol = int(rng.choice([0, 1, 2, 3, 5, 7, 20, 100]))
        sps = int(rng.choice([3, 8, 20, 100]))
        pt = int(rng.choice([0, 1, 8]))
        x = _signal(rng, n, mod, bps, oracle.noise_for_mod_type(mod))
        center = 0.0 if mod != "ASK" else 0.4
        # the model engine shards the demodulated signal, so "PSK" here only selects the -4.0 sentinel
        p = DemodParams(mod if mod != "PSK" else "FSK", bps, 0.0, center, 0.6, tol, sps, 0.1, pt, True)
        cuts = sorted(rng.choice(np.arange(1, n // 2), size=world - 1, replace=False) * 2) if n // 2 - 1 >= world - 1 else None
        if cuts is None:
            continue
        edges = [0] + [int(c) for c in cuts] + [n]
        bounds = [(edges[r], edges[r + 1]) for r in range(world)]
        shards = [x[a:b] for a, b in bounds]
        # every other case: the halo comes with the shard instead of through the first exchange
        halos = [None] + [torch.tensor([float(x[a - 1])], dtype=torch.float32) for a, _ in bounds[1:]] if it % 2 else None
        res = run_threads(world, lambda r: model_shard.ModelShardEngine(tile=int(rng.choice([16, 32])), span=8,
                                                                        chunk_tiles=int(rng.choice([1, 2]))),
                          shards, bounds, n, p, halos)
        assert_same(stitch(res), reference_result(oracle, x, p), (it, world, n, mod, bps, tol, sps, pt, bounds))


def test_shard_bounds():
    assert shard_bounds(8192, 2) == [(0, 4096), (4096, 8192)]
    b = shard_bounds(10_000, 4)
    assert b[0] == (0, 2560) and b[-1][1] == 10_000 and all(e - s >= 2 for s, e in b)
    with pytest.raises(ValueError):
        shard_bounds(5, 4)


def _gloo_worker(rank, world, port, n, seed, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        x = _signal(rng, n, "FSK", 1, np.float32(-4.0))
        x[n // 2 - 40:n // 2 + 25] = -4.0                      # a pause straddling the shard boundary
        p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 20, 0.1, 8, True)
        a, b = (0, n // 2) if rank == 0 else (n // 2, n)
        from urh_amd.sharding import RcclComm
        comm = RcclComm.create()                          # no RCCL on a gloo group: every rank agrees on torch.distributed's all-gathers
        assert isinstance(comm, TorchDistComm)
        pipe = ShardedPipeline(model_shard.ModelShardEngine(tile=32, span=8, chunk_tiles=2), comm)
        res = pipe.iq_to_bits(x[a:b], p, want_qad=True, pos_base=a, n_total=n)
        # the same with the halo handed over with the shard: two all-gathers instead of three
        res2 = pipe.iq_to_bits(x[a:b], p, want_qad=True, pos_base=a, n_total=n, halo_given=True,
                               left_halo=torch.tensor([float(x[a - 1])], dtype=torch.float32) if rank > 0 else None)
        q.put((rank, res, res2))
    finally:
        dist.destroy_process_group()


def test_sharded_protocol_over_gloo(oracle):
    """world_size 2, one process per rank, torch.distributed gloo: the N>1 path of bench.py on CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    n, seed, port = 3000, 5, 29500 + os.getpid() % 2000
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n, seed, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    items = [q.get(timeout=120) for _ in range(2)]
    got = {r: a for r, a, _ in items}
    got2 = {r: b for r, _, b in items}
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    rng = np.random.default_rng(seed)
    x = _signal(rng, n, "FSK", 1, np.float32(-4.0))
    x[n // 2 - 40:n // 2 + 25] = -4.0
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 20, 0.1, 8, True)
    assert_same(stitch([got[0], got[1]]), reference_result(oracle, x, p), "gloo")
    assert_same(stitch([got2[0], got2[1]]), reference_result(oracle, x, p), "gloo, halo with the shard")


class _FakeRccl:
    """stand-in for librccl.so (RcclComm.create's lib_loader hook): the five entry points, failing where told"""

    def __init__(self, fail_uid=False, fail_init=False, hang_init=False):
        import ctypes as C

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]
        self.UniqueId = UniqueId
        self.fail_uid, self.fail_init, self.hang_init = fail_uid, fail_init, hang_init
        self.destroyed = 0

    def ncclGetUniqueId(self, ref):
        return 5 if self.fail_uid else 0

    def ncclCommInitRank(self, ref, world, uid, rank):
        if self.hang_init:
            import time
            time.sleep(30)
        if self.fail_init:
            return 3
        ref._obj.value = 0x1234                              # (the communicator handle)
        return 0

    def ncclCommDestroy(self, comm):
        self.destroyed += 1
        return 0

    def ncclGetErrorString(self, rc):
        return b"fake"


def _fallback_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from urh_amd.sharding import RcclComm
        out = []
        cpu = torch.device("cpu")

        def attempt(tag, loader):
            comm = RcclComm.create(lib_loader=loader, flag_device=cpu)
            out.append((tag, type(comm).__name__, RcclComm.last_fallback_reason))
            # the group is still in step afterwards: one more collective that every rank must reach
            t = torch.tensor([rank + 1], dtype=torch.int32)
            dist.all_reduce(t)
            assert int(t.item()) == world * (world + 1) // 2

        def load_fails_on_rank_1():
            if rank == 1:
                raise OSError("librccl.so: cannot open shared object file")
            return _FakeRccl()
        attempt("load fails on rank 1", load_fails_on_rank_1)
        attempt("unique id fails on rank 0", lambda: _FakeRccl(fail_uid=(rank == 0)))
        lib_ok = _FakeRccl()
        attempt("init fails on rank 1", lambda: _FakeRccl(fail_init=True) if rank == 1 else lib_ok)
        out.append(("destroyed on the rank that had succeeded", lib_ok.destroyed if rank == 0 else None, None))
        RcclComm.INIT_TIMEOUT_S = 1.0
        attempt("init hangs on rank 0", lambda: _FakeRccl(hang_init=(rank == 0)))
        attempt("every step succeeds", lambda: _FakeRccl())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_rccl_comm_create_agrees_on_the_fallback():
    """RcclComm.create when ONE rank fails on its own -- the library does not load, rank 0 gets no unique id, ncclCommInitRank fails or
    does not return: every rank ends with a TorchDistComm (none is left waiting in a broadcast or a rendezvous) and says why; with every
    step succeeding on every rank it is an RcclComm.  (A stand-in library through create()'s lib_loader hook; flags on the CPU, gloo.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = 31500 + os.getpid() % 2000
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    items = dict(q.get(timeout=120) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank in (0, 1):
        res = {tag: (kind, why) for tag, kind, why in items[rank]}
        for tag in ("load fails on rank 1", "unique id fails on rank 0", "init fails on rank 1", "init hangs on rank 0"):
            assert res[tag][0] == "TorchDistComm" and res[tag][1], (rank, tag, res[tag])
        assert res["every step succeeds"] == ("RcclComm", None), (rank, res)
    assert dict((t, k) for t, k, _ in items[0])["destroyed on the rank that had succeeded"] == 1
    assert "cannot open" in dict((t, w) for t, _, w in items[1])["load fails on rank 1"]


class _FirEngine:
    """the two FIR entry points of an engine, counting; the filter is a running sum so the arithmetic is checkable"""

    def fir(self, iq, taps, hist):
        x = np.concatenate([np.zeros((len(taps) - 1, 2), np.float32) if hist is None else np.asarray(hist), np.asarray(iq)])
        m = len(taps)
        return torch.from_numpy(np.stack([x[i:i + m].sum(0) for i in range(len(iq))]).astype(np.float32))

    def fir_tail(self, iq, k):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(iq)[-k:]))


class _CountingComm:
    def __init__(self, rank, world):
        self.rank, self.world, self.calls = rank, world, 0

    def all_gather(self, t):
        self.calls += 1
        raise AssertionError("no collective may be entered in raw-halo mode")


def test_fir_filter_mode_is_the_same_on_every_rank():
    """rank 0 took "raw halo" from want_halo, ranks > 0 from left_raw -- with left_raw missing everywhere rank 0
    returned early and the others blocked in the all-gather.  The mode is one flag now: in raw mode NO rank enters a collective and a
    rank > 0 without left_raw raises; want_halo without raw mode raises on every rank."""
    m = 4
    taps = np.ones((m, 2), np.float32)
    x = torch.arange(40, dtype=torch.float32).reshape(20, 2)
    for rank in (0, 1):
        pipe = ShardedPipeline(_FirEngine(), _CountingComm(rank, 2))
        if rank == 0:
            out, halo = pipe.fir_filter(x[:10], taps, want_halo=True)
            assert halo is None and out.shape == (10, 2)
        else:
            with pytest.raises(ValueError, match="left_raw"):
                pipe.fir_filter(x[10:], taps, want_halo=True)
            with pytest.raises(ValueError, match="left_raw"):
                pipe.fir_filter(x[10:], taps, raw_halo=True)
            out, halo = pipe.fir_filter(x[10:], taps, left_raw=x[10 - (m + 1):10], want_halo=True)
            whole = _FirEngine().fir(x, taps, None)
            assert torch.equal(out, whole[10:]) and torch.equal(halo, whole[8:10])
        with pytest.raises(ValueError, match="raw_halo"):
            pipe.fir_filter(x[:10], taps, want_halo=True, raw_halo=False)
        assert pipe.comm.calls == 0
