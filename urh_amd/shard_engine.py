"""Per-rank GPU engine of the sharded IQ->bits pass: the four urhgpu_shard_* phases of liburhgpu.so
(include/urhgpu.h) behind the engine interface urh_amd/sharding.py orchestrates.  torch is plumbing
(device buffers, the stream, the tensors the all-gathers move); all arithmetic is in the HIP kernels."""
import ctypes as C

import numpy as np

from . import _lib
from .pipeline import DevicePipeline, _torch_dtype


class ShardResult:
    """This rank's piece of a sharded result (device tensors) + host views for `sharding.stitch`."""

    def __init__(self, qad, rows, bits, msg_off, pauses, pos, pos_off, counts, params, ctx=None):
        self.qad, self.rows_buf, self.bits_buf = qad, rows, bits
        self.msg_off_buf, self.pauses_buf, self.pos_buf, self.pos_off_buf = msg_off, pauses, pos, pos_off
        self.counts, self.params = counts, params
        self._host_counts = None
        self._ctx = ctx
        self._host = None

    def host_counts(self):
        if self._host_counts is None:
            if self._ctx is not None:
                self._ctx.join()
            c = self.counts.cpu().numpy()
            self._host_counts = tuple(int(x) for x in c[:4])
            self._rows_needed = int(c[4])
        return self._host_counts

    def check_capacity(self):
        n_rows, n_msg, n_bits, n_pos = self.host_counts()
        if self._rows_needed > self.rows_buf.shape[0] or n_msg > self.pauses_buf.shape[0] or n_bits > self.bits_buf.shape[0] \
                or (self.pos_buf is not None and n_pos > self.pos_buf.shape[0]):
            raise _lib.UrhGpuError(_lib.ERR_CAPACITY, f"output capacity too small: rows={self._rows_needed} msgs={n_msg} "
                                                      f"bits={n_bits} pos={n_pos}")

    def piece(self):
        """dict(rows, bits, msg_end, pauses, pos, pos_end) of numpy arrays (see sharding.stitch)."""
        self.check_capacity()
        n_rows, n_msg, n_bits, n_pos = self.host_counts()
        rows = self.rows_buf[:n_rows].cpu().numpy()
        if n_rows and rows[0, 0] == _lib.ROW_ABSORBED:
            rows = rows[1:]
        return dict(rows=rows, bits=self.bits_buf[:n_bits].cpu().numpy(),
                    msg_end=self.msg_off_buf[1:n_msg + 1].cpu().numpy(), pauses=self.pauses_buf[:n_msg].cpu().numpy(),
                    pos=self.pos_buf[:n_pos].cpu().numpy() if self.pos_buf is not None else np.zeros(0, np.int64),
                    pos_end=self.pos_off_buf[1:n_msg + 1].cpu().numpy())

    def host(self):
        """This rank's piece ON THE HOST from the compact blob (GpuShardEngine(host_results=True): packed at the end of the pass, copied
        by the copy stream while later passes run): a pipeline.HostBits whose views are valid until three passes later.  Rows, bits,
        pauses and offsets are the rank's LOCAL piece, as in `piece()`."""
        from .pipeline import HostBits
        if self._host is None:
            raise ValueError("the engine was not asked for host results (host_results=True)")
        if not isinstance(self._host, HostBits):
            eng, slot, copied, pass_no = self._host
            self._host = eng._finish_host_copy(slot, copied, self.params, self.qad, pass_no)
        return self._host

    # `stitch` accepts mappings: make the result itself usable as a piece
    def __getitem__(self, k):
        if not hasattr(self, "_piece"):
            self._piece = self.piece()
        return self._piece[k]


class GpuShardEngine(DevicePipeline):
    """DevicePipeline (context, stream, output buffers) + the shard phases.
    host_results=True: every pass also leaves its compact result blob (include/urhgpu.h) in pinned
    host memory -- packed at the end of the pass's tail, copied by a third stream while the following passes run, three blob slots in
    rotation, the copy sized by the previous pass's blob (what a prediction misses is fetched when the result is looked at):
    ShardResult.host().  The same window as the single-GPU CaptureStream, per rank."""

    def __init__(self, device: int = 0, pipelined: bool = False, tuning=None, tail_stream_priority: int = 0, host_results: bool = False,
                 worst_case_rows: bool = False):
        """worst_case_rows: size the pulse table for one row per (tolerance + 1) samples (a shard of noise) instead of the default
        four rows per symbol, which a pass that needs more reports as ERR_CAPACITY"""
        super().__init__(device, pipelined=pipelined, tuning=tuning, tail_stream_priority=tail_stream_priority)
        self.worst_case_rows = bool(worst_case_rows)
        self.host_results = bool(host_results)
        self._pass = 0
        self._hslots = None                                 # [(device blob, pinned host blob, copy-done event)] x 3
        self._copy_stream = None
        self._predicted = 0                                 # bytes the next copy is sized for (0: the whole blob)
        self._hqueued = [False, False, False]               # a copy has been queued for the slot (its event means something)
        self._slot_pass = [-1, -1, -1]                      # the pass whose blob the slot holds: a result looked at too late says so

    def _host_slot(self, cap_rows, cap_bits, cap_msg, cap_pos, has_pos):
        torch = self.torch
        cap = int(_lib.load().urhgpu_blob_capacity(cap_rows, cap_bits, cap_msg, cap_pos, 1 if has_pos else 0))
        if self._hslots is None or self._hslots[0][0].numel() < cap:
            self._hslots = [(torch.empty(cap, dtype=torch.uint8, device=self.device), torch.empty(cap, dtype=torch.uint8).pin_memory(),
                             torch.cuda.Event()) for _ in range(3)]
            self._hevents = [torch.cuda.Event() for _ in range(3)]     # "the pass's tail has packed the blob" (one per slot: no event made per pass)
            self._copy_stream = torch.cuda.Stream(self.device)
            self._predicted = 0
            self._hqueued = [False, False, False]
            self._slot_pass = [-1, -1, -1]                  # (results of earlier passes point at the old buffers: host() on them raises)
        k = self._pass % 3
        self._slot_pass[k] = self._pass
        self._pass += 1
        return k, cap

    def _queue_host_copy(self, k):
        """behind the pass's tail (the current stream), on the copy stream: the first `predicted` bytes of the blob"""
        torch = self.torch
        dblob, hblob, done = self._hslots[k]
        if self._predicted <= 0:
            # nobody has looked at a result yet: size the copy by whichever earlier blob has already arrived (else: the whole slot)
            for j in range(3):
                if j != k and self._hqueued[j] and self._hslots[j][2].query():
                    hdr = self._hslots[j][1][:128].numpy().view(np.int64)
                    if int(hdr[0]) == _lib.BLOB_MAGIC:
                        self._predicted = max(self._predicted, abs(int(hdr[6])) + abs(int(hdr[6])) // 8 + 65536)
        n = dblob.numel() if self._predicted <= 0 else min(dblob.numel(), self._predicted)
        self._hqueued[k] = True
        ev = self._hevents[k]
        ev.record(torch.cuda.current_stream(self.device))
        self._copy_stream.wait_event(ev)
        with torch.cuda.stream(self._copy_stream):
            hblob[:n].copy_(dblob[:n], non_blocking=True)
            done.record(self._copy_stream)
        return n

    def _finish_host_copy(self, k, copied, params, qad, pass_no):
        from .pipeline import HostBits
        if self._slot_pass[k] != pass_no:
            raise RuntimeError(f"ShardResult.host(): the blob slot of pass {pass_no} has been reused by pass {self._slot_pass[k]} (three blob slots "
                               f"rotate: look at a result before three later passes have been issued) or the slots were reallocated")
        dblob, hblob, done = self._hslots[k]
        done.synchronize()
        hdr = hblob[:128].numpy().view(np.int64)
        total = abs(int(hdr[6]))
        if total > copied:                                   # the prediction fell short: the rest now
            with self.torch.cuda.stream(self._copy_stream):
                hblob[copied:total].copy_(dblob[copied:total], non_blocking=False)
        self._predicted = total + total // 8 + 65536
        h = self._host_bits(HostBits, hblob, params, qad)
        h.sharded_piece = True
        return h

    @staticmethod
    def _host_bits(HostBits, hblob, params, qad):
        return HostBits.from_blob(hblob.data_ptr(), params, n_samples=int(qad.shape[0]) if qad is not None else 0,
                                  d_qad=qad.data_ptr() if qad is not None else 0)

    def tail(self, iq_local, p):
        torch = self.torch
        if iq_local.dtype == torch.complex64:
            iq_local = torch.view_as_real(iq_local)
        return iq_local[-2:].contiguous()                     # (2, 2): samples n-2, n-1

    def halo_view(self, left_halo):
        """(2, 2) view of a caller-provided halo (complex64 (2,) or (2, 2) in the capture's dtype)"""
        torch = self.torch
        if left_halo.dtype == torch.complex64:
            left_halo = torch.view_as_real(left_halo)
        if tuple(left_halo.shape) != (2, 2):
            raise ValueError("left_halo: the two samples before the shard, shape (2, 2)")
        return left_halo.contiguous()

    def fir_tail(self, iq_local, k):
        torch = self.torch
        if iq_local.dtype == torch.complex64:
            iq_local = torch.view_as_real(iq_local)
        return iq_local[-k:].contiguous()

    def fir(self, iq_local, taps, left):
        """complex64 FIR of this shard (float32 (N, 2) or complex64 (N,)) with `left` (the m-1 preceding samples) as history"""
        torch = self.torch
        x = torch.view_as_real(iq_local) if iq_local.dtype == torch.complex64 else iq_local
        h = torch.view_as_real(taps) if taps.dtype == torch.complex64 else taps
        if x.dtype != torch.float32 or h.dtype != torch.float32 or not x.is_contiguous():
            raise ValueError("FIR needs contiguous float32 / complex64 samples and taps")
        if h.dim() != 2 or h.shape[1] != 2:
            raise ValueError("taps: complex64 (m,) or float32 (m, 2)")          # a flat float32 vector would be read as twice the taps
        h = h.contiguous()
        out = torch.empty_like(x)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_fir_filter_dev(self.ctx.handle, C.c_void_p(x.data_ptr()), x.shape[0], C.c_void_p(h.data_ptr()),
                                                     h.shape[0], C.c_void_p(left.data_ptr()) if left is not None else None,
                                                     C.c_void_p(out.data_ptr())))
        self._fir_keep = (x, h, left)
        return out if iq_local.dtype != torch.complex64 else torch.view_as_complex(out)

    def _setup(self, iq, p, want_qad, n_total=0):
        torch = self.torch
        if iq.dtype == torch.complex64:
            iq = torch.view_as_real(iq)
        if iq.dim() != 2 or iq.shape[1] != 2 or not iq.is_contiguous():
            raise ValueError("IQ must be a contiguous (N, 2) tensor")
        npdt = _torch_dtype(iq)
        n = iq.shape[0]
        cp = p.to_c(npdt)
        cap_rows, cap_bits, cap_msg, cap_pos = self.capacities(n, p, (n // (p.tolerance + 1) + 2) if self.worst_case_rows else None)
        # a rank owns the rows that END in its shard -- one that began in an earlier shard (a stuck carrier across the boundary) -- and, ASK,
        # the equal-state rows of LATER shards merged into its last one: a row can span the whole capture, whatever the shard's size
        extra = (max(int(n_total) - n, 0) // max(int(p.samples_per_symbol), 1) + 1) * int(p.bits_per_symbol)
        cap_bits += extra
        cap_pos += extra
        qad = self._buf("qad", (n,), torch.float32) if want_qad else None
        rows = self._buf("rows", (cap_rows, 2), torch.int64)
        bits = self._buf("bits", (cap_bits,), torch.uint8)
        msg_off = self._buf("msg_off", (cap_msg + 1,), torch.int64)
        pauses = self._buf("pauses", (cap_msg,), torch.int64)
        pos_off = self._buf("pos_off", (cap_msg + 1,), torch.int64)
        pos = self._buf("pos", (cap_pos,), torch.int64) if p.write_bit_sample_pos else None
        counts = self._buf("counts", (5,), torch.int64)
        o = _lib.Outputs()
        o.qad = qad.data_ptr() if qad is not None else None
        o.rows = rows.data_ptr(); o.cap_rows = cap_rows
        o.bits = bits.data_ptr(); o.cap_bits = cap_bits
        o.msg_off = msg_off.data_ptr(); o.pauses = pauses.data_ptr(); o.cap_msg = cap_msg
        o.pos = pos.data_ptr() if pos is not None else None
        o.cap_pos = cap_pos if pos is not None else 0
        o.pos_off = pos_off.data_ptr()
        o.counts = counts.data_ptr()
        self._hslot = None
        if self.host_results:
            k, cap = self._host_slot(cap_rows, cap_bits, cap_msg, cap_pos if pos is not None else 0, pos is not None)
            dblob, _, done = self._hslots[k]
            # the blob slot is written by this pass's tail: behind the copy that last read it (three passes ago)
            (self.tail_stream if self.tail_stream is not None else torch.cuda.current_stream(self.device)).wait_event(done)
            o.blob = dblob.data_ptr(); o.cap_blob = cap
            self._hslot = k
        self._res = ShardResult(qad, rows, bits, msg_off, pauses, pos, pos_off, counts, p, self.ctx)
        self._ask = p.modulation_type == "ASK"
        self._keep = (iq,)                                     # keep the inputs alive until the pass is over
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        return iq, n, cp, o

    def runs_begin(self, iq, pos_base, n_total, rank, world, p, want_qad):
        """start the hot kernel on every chunk but the first while the halo all-gather is still in flight"""
        iq, n, cp, o = self._setup(iq, p, want_qad, n_total)
        self._pre = (iq, n, cp, o)
        _lib.check(_lib.load().urhgpu_shard_prelaunch_dev(self.ctx.handle, C.c_void_p(iq.data_ptr()), n, int(pos_base), int(n_total),
                                                          int(rank), int(world), C.byref(cp), C.byref(o)))

    def runs_launch(self, iq, left, pos_base, n_total, rank, world, p, want_qad):
        """the whole hot launch: the halo came with the shard (no exchange)"""
        iq, n, cp, o = self._setup(iq, p, want_qad, n_total)
        self._pre = (iq, n, cp, o)
        self._keep += (left,)
        _lib.check(_lib.load().urhgpu_shard_launch_dev(self.ctx.handle, C.c_void_p(iq.data_ptr()), n, int(pos_base), int(n_total),
                                                       int(rank), int(world), C.c_void_p(left.data_ptr()) if left is not None else None,
                                                       C.byref(cp), C.byref(o)))

    def runs(self, iq, left, pos_base, n_total, rank, world, p, want_qad):
        torch = self.torch
        pre = getattr(self, "_pre", None)
        if pre is not None:
            iq, n, cp, o = pre
            self._pre = None
        else:
            iq, n, cp, o = self._setup(iq, p, want_qad, n_total)
        self._keep += (left,)
        if left is not None and getattr(self, "tail_stream", None) is not None:
            left.record_stream(self.tail_stream)                # gathered under the caller's stream, read by the first chunk on the tail stream
        summary = self._buf("summary", (9,), torch.int64)      # URHGPU_SHARD_SUMMARY_BYTES = 72
        lh = C.c_void_p(left.data_ptr()) if left is not None else None
        _lib.check(_lib.load().urhgpu_shard_runs_dev(self.ctx.handle, C.c_void_p(iq.data_ptr()), n, int(pos_base), int(n_total),
                                                     int(rank), int(world), lh, C.byref(cp), C.byref(o),
                                                     C.c_void_p(summary.data_ptr())))
        return summary

    def rows(self, summaries):
        torch = self.torch
        self._keep += (summaries,)
        merge = self._buf("merge", (5,), torch.int64) if self._ask else None
        _lib.check(_lib.load().urhgpu_shard_rows_dev(self.ctx.handle, C.c_void_p(summaries.data_ptr()),
                                                     C.c_void_p(merge.data_ptr()) if merge is not None else None))
        return merge

    def bits_prepare(self, merged_all):
        torch = self.torch
        self._keep += (merged_all,)
        flags = self._buf("flags", (3,), torch.int64)
        _lib.check(_lib.load().urhgpu_shard_bits_prepare_dev(
            self.ctx.handle, C.c_void_p(merged_all.data_ptr()) if merged_all is not None else None, C.c_void_p(flags.data_ptr())))
        return flags

    def bits_finish(self, flags_all):
        self._keep += (flags_all,)
        _lib.check(_lib.load().urhgpu_shard_bits_finish_dev(self.ctx.handle, C.c_void_p(flags_all.data_ptr())))
        if self._hslot is not None:
            self._res._host = (self, self._hslot, self._queue_host_copy(self._hslot), self._slot_pass[self._hslot])
        res, self._res = self._res, None
        res._keep = self._keep
        self._keep = ()
        return res
