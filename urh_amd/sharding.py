"""Sample-contiguous sharding of one long capture over the GPUs of a node.

Rank r holds samples [r*n_local, (r+1)*n_local) of the capture (the last rank may hold more or
fewer).  The hot kernel runs on every shard independently; what crosses a shard boundary is tiny and
is exchanged with three (ASK: four) small all-gathers -- no data-path collective ever moves samples:

    1. halo        the last two IQ samples of every shard (16 B)      -> seam of the FSK conj-product and
                                                                        the state of the sample before the shard
    2. summary     one ChunkInfo (72 B) per shard: the shard's run structure reduced to what its
                   neighbours need (leading run length, first / last stable run, still-short trailing run,
                   number of accepted runs)                           -> pulse-table rows with global lengths
    3. (ASK only)  first / last row of every shard's merged table     -> equal-state rows merged across shards
    4. bits        three flags per shard (long pause present, data before the first / after the last one)
                                                                      -> which boundary-spanning groups are messages

The result stays sharded: rank r owns the pulse-table rows that END in its shard and the bits /
bit_sample_pos those rows expand to; concatenating the ranks' pieces in rank order gives exactly the
single-GPU (and reference) result (`stitch`).  Costas/PSK carries loop state across the whole capture and
does not shard ("replicas only").

The orchestration below is engine-agnostic: `engine` is the GPU engine (urh_amd.shard_engine.GpuShardEngine,
HIP kernels behind the C ABI) in production; the CPU test-suite drives the same orchestration with the executable
model of the kernels (tests/model_shard.py) over a world_size-2 gloo group.
"""
import contextlib

import numpy as np


class TorchDistComm:
    """all_gather over torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather(self, t):
        """t: torch tensor (same shape/dtype on every rank) -> tensor [world, *t.shape] on t's device."""
        return self.all_gather_start(t)()

    def all_gather_start(self, t):
        """Enqueue the all-gather and return a function that waits for it (stream-level on GPU tensors: the host does not
        block) and returns the gathered tensor: work enqueued in between overlaps the collective."""
        import torch
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # GPU tensors over a gloo group (two ranks sharing ONE GPU, which RCCL refuses: tools/two_ranks_one_gpu.sh): through the host
            torch.cuda.current_stream(t.device).synchronize()
            host = t.contiguous().cpu()
            out_h = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype)
            self.dist.all_gather_into_tensor(out_h.view(-1).view(torch.uint8), host.view(-1).view(torch.uint8), group=self.group)
            out_d = out_h.to(t.device)
            return lambda: out_d
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # gathered as raw bytes: halos of uint16 captures are torch.uint16 tensors, which the nccl / gloo backends do not all take
        work = self.dist.all_gather_into_tensor(out.view(-1).view(torch.uint8), t.contiguous().view(-1).view(torch.uint8),
                                                group=self.group, async_op=True)

        def wait():
            work.wait()
            return out
        return wait


class RcclComm:
    """all_gather straight through RCCL: ncclAllGather (ctypes on the librccl.so torch itself has loaded) enqueued on the CURRENT
    torch stream with a communicator of its own -- a few microseconds of host time per exchange and no hop through a collective
    stream, against 55-80 us per torch.distributed all_gather_into_tensor (round 3, profiles/HISTORY.md: three of those per pass made
    the sharded pass host-bound).  The communicator's unique id travels through the torch.distributed group once, at start-up;
    torch.distributed is still what launches and synchronises the ranks.  `RcclComm.create(group)` falls back to TorchDistComm when
    the library or the communicator cannot be had -- on EVERY rank or on none: each step's outcome is agreed on through the group
    before the next collective step begins (`last_fallback_reason` says which step failed)."""

    _UID_BYTES = 128
    INIT_TIMEOUT_S = 120.0                 # ncclCommInitRank that has not returned by then counts as failed (the rank falls back with the others)
    last_fallback_reason = None            # why the last create() returned a TorchDistComm (None: it returned an RcclComm)

    @staticmethod
    def load_library(lib_path=None):
        """librccl.so beside torch with the five entry points this class calls, prototypes set (raises when it cannot be had)"""
        import ctypes as C
        import os
        import torch
        if lib_path is None:
            lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = C.CDLL(lib_path)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * RcclComm._UID_BYTES)]
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.restype = C.c_int
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllGather.restype = C.c_int
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.restype = C.c_int
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.UniqueId = UniqueId
        return lib

    def __init__(self, group=None, lib_path=None):
        """every step on every rank succeeds or raises: use create() when a rank may fail on its own"""
        import torch
        comm, reason = self._create(group, lambda: self.load_library(lib_path), torch.device("cuda", torch.cuda.current_device()), into=self)
        if comm is None:
            raise RuntimeError(reason)

    @classmethod
    def create(cls, group=None, lib_loader=None, flag_device=None):
        """RcclComm, or TorchDistComm when RCCL cannot be reached directly (no GPU, no librccl.so beside torch, the unique id or the
        communicator could not be made, ncclCommInitRank did not return within INIT_TIMEOUT_S).  Every rank takes the same branch:
        after each step that a rank can fail on its own -- loading the library, rank 0's ncclGetUniqueId, ncclCommInitRank -- the
        ranks agree on the outcome (all_reduce MIN over the group) BEFORE any of them enters the next collective step; a rank that
        failed early therefore never leaves the others waiting in a broadcast or a communicator rendezvous.
        lib_loader / flag_device: test hooks (a stand-in library; flags on the CPU under gloo)."""
        import torch
        import torch.distributed as dist
        on_gpu = torch.cuda.is_available() and dist.get_backend(group) == "nccl"
        dev = flag_device if flag_device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
        if lib_loader is None:
            def lib_loader():
                if not on_gpu:
                    raise RuntimeError("not an RCCL process group (no GPU, or the group's backend is not nccl)")
                return cls.load_library()
        comm, reason = cls._create(group, lib_loader, dev)
        cls.last_fallback_reason = reason
        return comm if comm is not None else TorchDistComm(group)

    @classmethod
    def _create(cls, group, lib_loader, dev, into=None):
        import ctypes as C
        import threading
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        src = dist.get_global_rank(group, 0) if group is not None else 0

        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1

        # step 1 (every rank on its own): the library and its symbols
        lib, err = None, None
        try:
            lib = lib_loader()
        except Exception as exc:                              # noqa: BLE001 -- any failure means "use torch.distributed"
            err = f"rank {rank}: librccl.so not usable: {exc!r}"
        if not agree(lib is not None):
            return None, err or "another rank could not load librccl.so"
        # step 2 (rank 0 on its own, then one broadcast every rank takes part in): the unique id, with rank 0's verdict in front
        uid = lib.UniqueId()
        ok0 = 1
        if rank == 0:
            try:
                rc = lib.ncclGetUniqueId(C.byref(uid))
                if rc != 0:
                    ok0, err = 0, f"ncclGetUniqueId: {rc}"
            except Exception as exc:                          # noqa: BLE001
                ok0, err = 0, f"ncclGetUniqueId: {exc!r}"
        t = torch.frombuffer(bytearray(bytes([ok0]) + bytes(uid)), dtype=torch.uint8).to(dev)
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[0] != 1:
            return None, err or "rank 0 could not make a unique id"
        uid = lib.UniqueId.from_buffer_copy(raw[1:])
        # step 3 (a rendezvous of the ranks inside RCCL): bounded by a timeout, and agreed on afterwards
        comm = C.c_void_p()
        box = {}

        # The communicator binds to the CALLING THREAD's current device, and a new thread starts on device 0 whatever the process has
        # selected: without the set_device below every rank of a node with all GPUs visible would offer device 0 to RCCL
        # (ncclInvalidUsage, the same refusal as two ranks on one GPU -- profiles/r05_rccl_two_ranks_one_gpu.txt).
        cur_dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def init():
            try:
                if cur_dev is not None:
                    torch.cuda.set_device(cur_dev)
                box["rc"] = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
            except Exception as exc:                          # noqa: BLE001
                box["exc"] = exc
            if box.get("abandoned") and box.get("rc") == 0:   # it came up after the timeout, when every rank had fallen back: nobody will use it
                try:
                    lib.ncclCommDestroy(comm)
                except Exception:                             # noqa: BLE001
                    pass
        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(cls.INIT_TIMEOUT_S)
        if th.is_alive():
            box["abandoned"] = True
        ok = (not th.is_alive()) and box.get("rc") == 0
        if not ok:
            err = (f"rank {rank}: ncclCommInitRank did not return within {cls.INIT_TIMEOUT_S:.0f} s" if th.is_alive() else
                   f"rank {rank}: ncclCommInitRank: {box.get('exc') or box.get('rc')}")
        self = into if into is not None else cls.__new__(cls)
        self.torch, self.C, self.lib = torch, C, lib
        self.rank, self.world = rank, world
        self.comm = comm if ok else None
        if not agree(ok):
            if ok:
                self.close()
            return None, err or "another rank could not join the communicator"
        return self, None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()} ({rc})")

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    def __repr__(self):
        return f"RcclComm(rank={self.rank}, world={self.world})"

    def _gather(self, t, stream):
        torch = self.torch
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        nbytes = t.numel() * t.element_size()
        self._check(self.lib.ncclAllGather(t.data_ptr(), out.data_ptr(), nbytes, 1, self.comm, stream.cuda_stream), "ncclAllGather")   # 1 = ncclUint8
        return t, out

    def all_gather(self, t):
        """on the current stream, in order with the kernels around it (no event, no other stream)"""
        torch = self.torch
        with torch.cuda.device(t.device):
            keep, out = self._gather(t, torch.cuda.current_stream(t.device))
        out._urh_keep = keep                                  # the send buffer lives as long as the result
        return out

    def all_gather_start(self, t):
        """the same, as a function to call for the result (the engine-agnostic orchestration starts the halo exchange before the hot
        kernel and picks the result up after it): enqueued right here, on the current stream"""
        out = self.all_gather(t)
        return lambda: out


class ThreadComm:
    """W ranks as W threads of one process (lock-step through a barrier): lets a 1-GPU box (and plain CPU
    tests) execute the sharded path for any world size."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def all_gather_start(self, t):
        out = self.all_gather(t)
        return lambda: out

    def all_gather(self, t):
        import torch
        sh = self.shared
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        sh.slots[self.rank] = t
        sh.barrier.wait()
        out = torch.stack([s.to(t.device) for s in sh.slots])
        if out.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
        sh.barrier.wait()
        return out


def shard_bounds(n_total: int, world: int):
    """[begin, end) of every rank's shard: equal shards of ceil(n/world) samples rounded up to a multiple of
    64 (so that every shard starts 16-byte aligned for every IQ dtype), the last rank takes what is left.
    Every shard needs >= 2 samples."""
    per = -(-n_total // world)
    per = -(-per // 64) * 64
    b = [min(r * per, n_total) for r in range(world + 1)]
    b[world] = n_total
    if any(b[r + 1] - b[r] < 2 for r in range(world)):
        raise ValueError(f"capture of {n_total} samples is too short to shard over {world} ranks")
    return [(b[r], b[r + 1]) for r in range(world)]


class ShardedPipeline:
    """One rank's view of the sharded IQ->bits pass."""

    def __init__(self, engine, comm):
        self.engine, self.comm = engine, comm
        self.rank, self.world = comm.rank, comm.world

    # bench.py / DevicePipeline compatible surface ------------------------------------------------
    @property
    def ctx(self):
        return self.engine.ctx

    def reserve(self, n_local, p):
        self.engine.reserve(n_local, p)

    def fir_filter(self, iq_local, taps, left_raw=None, want_halo=False, raw_halo=None):
        """Signal.filter_range semantics on a sharded capture (BASELINE.json configs[3], "FIR-halo exchange"): every rank
        filters its shard with the m-1 samples that precede it as history; rank 0 starts from zero history like the
        reference's fir_filter (signal_functions.pyx:513-525).  Returns the filtered shard (same shape as iq_local).
        left_raw is None: the history is the left neighbour's tail, one all-gather of (m-1) * 8 bytes per rank.
        left_raw given (every rank but the first; round 5): whoever distributed the capture handed the rank the m + 1 RAW samples
        that precede its shard ((m + 1, 2) float32 / complex64 (m + 1,): 520 bytes for 64 taps) -- no exchange at all: the last
        m - 1 of them are the filter's history, and filtering the m + 1 themselves gives, in their last two outputs, the two FILTERED
        samples before the shard, i.e. the demodulation's halo (want_halo=True: returns (filtered shard, that halo or None)).
        raw_halo states the mode and must be THE SAME ON EVERY RANK (a rank that guessed it from its own arguments could skip a
        collective the others enter): True = raw mode (rank 0 passes no left_raw, every other rank must), False = exchange,
        None = raw mode iff want_halo (the halo only exists in raw mode; left_raw without want_halo also selects it on ranks > 0,
        where rank 0 then has to say raw_halo=True)."""
        e, c = self.engine, self.comm
        if raw_halo is None:
            raw_halo = bool(want_halo) or left_raw is not None
        m = int(taps.shape[0])                     # complex64 (m,) or float32 (m, 2): rows = taps
        if m <= 1 or self.world == 1:
            out = e.fir(iq_local, taps, None)
            return (out, None) if want_halo else out
        if int(iq_local.shape[0]) < m - 1:
            raise ValueError("shard shorter than the filter history")
        if want_halo and not raw_halo:
            raise ValueError("want_halo needs raw_halo: the filtered halo comes from the raw samples handed over with the shard")
        if raw_halo:
            if self.rank == 0:
                out = e.fir(iq_local, taps, None)
                return (out, None) if want_halo else out
            if left_raw is None:
                raise ValueError("raw_halo: ranks > 0 pass the m + 1 raw samples that precede their shard as left_raw")
            raw = left_raw
            if hasattr(e, "torch") and raw.dtype == e.torch.complex64:
                raw = e.torch.view_as_real(raw)
            if int(raw.shape[0]) != m + 1:
                raise ValueError("left_raw: the m + 1 raw samples that precede the shard")
            out = e.fir(iq_local, taps, raw[2:].contiguous())
            if not want_halo:
                return out
            return out, e.fir(raw.contiguous(), taps, None)[-2:].contiguous()     # outputs m - 1 and m have their full history inside `raw`
        tails = c.all_gather(e.fir_tail(iq_local, m - 1))
        return e.fir(iq_local, taps, tails[self.rank - 1] if self.rank > 0 else None)

    def iq_to_bits(self, iq_local, p, want_qad=True, pos_base=None, n_total=None, halo_given=False, left_halo=None):
        """iq_local: this rank's shard.  pos_base / n_total default to equal shards of len(iq_local).
        halo_given (the same on every rank): whoever distributed the capture handed every rank but the first the two samples that
        precede its shard (left_halo: (2, 2) in the shard's dtype, or complex64 (2,)) -- 16 bytes more per rank to read from the
        file.  The halo exchange is then skipped: two all-gathers per pass (ASK: three) instead of three (four)."""
        e, c = self.engine, self.comm
        n_local = int(iq_local.shape[0])
        if pos_base is None:
            pos_base = self.rank * n_local
        if n_total is None:
            n_total = self.world * n_local
        if p.modulation_type == "PSK":
            raise ValueError("the Costas loop carries state across the whole capture: PSK does not shard")
        if halo_given and self.rank > 0 and left_halo is None:
            raise ValueError("halo_given: ranks > 0 pass the two samples before their shard as left_halo")
        pending = left = None
        if not halo_given:
            # the halo exchange overlaps the hot kernel (all chunks but the first); on a pipelined engine it is issued on the tail
            # stream (behind the previous pass's exchanges), which first waits for whatever produced the shard on the caller's stream
            with (e.halo_context() if hasattr(e, "halo_context") else contextlib.nullcontext()):
                pending = c.all_gather_start(e.tail(iq_local, p))
            if hasattr(e, "runs_begin"):
                e.runs_begin(iq_local, pos_base, n_total, self.rank, self.world, p, want_qad)
        else:
            if self.rank > 0:
                left = e.halo_view(left_halo) if hasattr(e, "halo_view") else left_halo
            if hasattr(e, "runs_launch"):          # the whole hot launch, on the hot stream
                e.runs_launch(iq_local, left, pos_base, n_total, self.rank, self.world, p, want_qad)
        # Everything after the hot kernel -- the wait for the halo, the first chunk, the all-gathers -- is issued on the engine's tail
        # stream when it is pipelined: the next pass's hot kernel then overlaps this pass's latency-bound tail, and the hot stream
        # never waits for a collective (the exchanges of one process group run in issue order: the halo of pass i + 1 sits behind the
        # last exchange of pass i's tail, so a hot stream that waited for its halo would run in lock-step with the tails).
        with (e.tail_context() if hasattr(e, "tail_context") else contextlib.nullcontext()):
            if pending is not None:
                halos = pending()
                left = halos[self.rank - 1] if self.rank > 0 else None
            summary = e.runs(iq_local, left, pos_base, n_total, self.rank, self.world, p, want_qad)
            merge = e.rows(c.all_gather(summary))
            merged_all = c.all_gather(merge) if merge is not None else None
            flags = e.bits_prepare(merged_all)
            return e.bits_finish(c.all_gather(flags))


def stitch(pieces):
    """Concatenate the per-rank pieces (rank order) of a sharded result into the single-GPU / reference
    shaped flat result.  pieces[r] = dict(rows, bits, msg_end, pauses, pos, pos_end) of numpy arrays, where
    msg_end / pos_end are LOCAL end offsets (in that rank's bits / pos) of the messages that close on rank r.
    Returns (ppseq, bits, msg_off, pauses, pos, pos_off)."""
    rows = np.concatenate([np.asarray(p["rows"], dtype=np.int64).reshape(-1, 2) for p in pieces])
    bits = np.concatenate([np.asarray(p["bits"], dtype=np.uint8) for p in pieces])
    pos = np.concatenate([np.asarray(p["pos"], dtype=np.int64) for p in pieces])
    pauses = np.concatenate([np.asarray(p["pauses"], dtype=np.int64) for p in pieces])
    msg_off, pos_off = [0], [0]
    b0 = p0 = 0
    for p in pieces:
        msg_off.extend((b0 + np.asarray(p["msg_end"], dtype=np.int64)).tolist())
        pos_off.extend((p0 + np.asarray(p["pos_end"], dtype=np.int64)).tolist())
        b0 += len(p["bits"])
        p0 += len(p["pos"])
    return rows, bits, np.array(msg_off, np.int64), pauses, pos, np.array(pos_off, np.int64)
