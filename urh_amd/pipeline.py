"""Device-resident IQ -> bits pipeline: the capture stays in HBM, one call produces the demodulated
signal (Signal.qad), the pulse table (grab_pulse_lens) and bits / pauses / bit_sample_pos
(ProtocolAnalyzer._ppseq_to_bits) without touching the host.

PyTorch is plumbing only (device allocations, streams, torch.distributed for sharded captures):
every byte of arithmetic happens in liburhgpu.so behind the C ABI (include/urhgpu.h).

Parameter names/defaults follow the reference's Signal object
(/root/reference/src/urh/signalprocessing/Signal.py:42-109).
"""
import array
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .signal_functions import dtype_code, mod_code


@dataclass
class DemodParams:
    modulation_type: str = "FSK"          # Signal.modulation_type
    bits_per_symbol: int = 1
    noise_threshold: float = 0.0
    center: float = 0.02
    center_spacing: float = 1.0
    tolerance: int = 5
    samples_per_symbol: int = 100
    costas_loop_bandwidth: float = 0.1
    pause_threshold: int = 8
    write_bit_sample_pos: bool = True

    def to_c(self, dtype, demod_only: bool = False) -> _lib.Params:
        """demod_only: for afp_demod alone -- the slicing parameters (tolerance, samples_per_symbol, bits_per_symbol as a symbol
        width) play no part there and must not make it fail (the reference's qad depends on the demodulation parameters only)."""
        if demod_only:
            from dataclasses import replace
            return replace(self, tolerance=0, samples_per_symbol=1, bits_per_symbol=max(1, int(self.bits_per_symbol))).to_c(dtype)
        mod, sentinel = mod_code(self.modulation_type)
        # the reference's own failures for these inputs: `uint16 tolerance` (signal_functions.pyx:392) raises
        # OverflowError, `n / samples_per_symbol` (ProtocolAnalyzer.py:353) ZeroDivisionError
        if not 0 <= int(self.tolerance) <= 65535:
            raise OverflowError("tolerance must fit an unsigned 16-bit integer")
        if int(self.samples_per_symbol) < 1:
            raise ZeroDivisionError("samples_per_symbol must be at least 1")
        if int(self.bits_per_symbol) < 1:
            raise ValueError("bits_per_symbol must be at least 1")
        p = _lib.Params()
        p.dtype = dtype_code(dtype)
        p.mod = mod
        p.bits_per_symbol = int(self.bits_per_symbol)
        p.noise_threshold = float(self.noise_threshold)
        p.center = float(self.center)
        p.center_spacing = float(self.center_spacing)
        p.tolerance = int(self.tolerance)
        p.samples_per_symbol = int(self.samples_per_symbol)
        p.costas_loop_bandwidth = float(self.costas_loop_bandwidth)
        p.pause_threshold = int(self.pause_threshold)
        p.write_bit_sample_pos = 1 if self.write_bit_sample_pos else 0
        p.noise_other = float(sentinel)
        p.mod_order = 0
        return p


_TORCH_DT = None
_PINNED_START = 1 << 20            # first size of a pinned result buffer (grown to the blob's real size on demand)


def _torch_dtype(t):
    import torch
    m = {torch.int8: np.int8, torch.uint8: np.uint8, torch.int16: np.int16, torch.float32: np.float32}
    if hasattr(torch, "uint16"):
        m[torch.uint16] = np.uint16
    if t.dtype not in m:
        raise ValueError("Unsupported dtype")
    return np.dtype(m[t.dtype])


class BitsResult:
    """Device-resident outputs of one IQ->bits pass (torch tensors) + lazy host views."""

    def __init__(self, qad, rows, bits, msg_off, pauses, pos, pos_off, counts, params, ctx=None, pipe=None, outputs=None):
        self.qad, self.rows_buf, self.bits_buf = qad, rows, bits
        self.msg_off_buf, self.pauses_buf, self.pos_buf, self.pos_off_buf = msg_off, pauses, pos, pos_off
        self.counts = counts
        self.params = params
        self._host_counts = None
        self._ctx = ctx
        self._pipe, self._outputs = pipe, outputs          # the pipeline and the C descriptor of the pass: host() packs through them
        self._hostbits = None

    def host(self, pool=None) -> "HostBits":
        """The results ON THE HOST through the compact blob (include/urhgpu.h): one more kernel packs them (5 B per pulse-table row, one bit
        per bit, 4 B per position), ONE copy into pinned memory moves them -- instead of five synchronous pageable copies of the wide int64
        tables (28 ms for a million rows; this: well under a millisecond).  pool: a dict owned by the caller that keeps the pinned buffer
        (the HostBits views it: valid until the next host() with the same pool); default: a buffer of the pipeline, valid until ANY
        result's next host() call -- the buffer is stamped with its current owner, and a result whose views were overwritten by another
        result's host() packs again instead of handing out a stale cache (ppseq() / flat() / messages() return copies).
        The pinned buffer follows the blob's REAL size (header first), not its capacity -- a pulse table's capacity is the worst case
        n / (tolerance + 1) rows, hundreds of MB for a 1 GiB capture whose blob is a few MB."""
        if self._pipe is None or self._outputs is None:
            raise ValueError("this result was not made by a DevicePipeline pass")
        pipe, o = self._pipe, self._outputs
        keep = pipe._pinned if pool is None else pool
        if self._hostbits is not None and pool is None and keep.get("owner") is self._hostbits:
            return self._hostbits
        torch = pipe.torch
        has_pos = self.pos_buf is not None
        cap = int(_lib.load().urhgpu_blob_capacity(int(o.cap_rows), int(o.cap_bits), int(o.cap_msg), int(o.cap_pos), 1 if has_pos else 0))
        dblob = pipe._buf("blob", (cap,), torch.uint8)
        o2 = _lib.Outputs()
        C.memmove(C.byref(o2), C.byref(o), C.sizeof(_lib.Outputs))
        o2.blob = dblob.data_ptr(); o2.cap_blob = cap
        total = C.c_int64(0)
        pipe.ctx.set_stream(torch.cuda.current_stream(pipe.device).cuda_stream)
        keep["owner"] = None                                  # whatever viewed the buffer is stale from here on
        hbuf = keep.get("blob")
        if hbuf is None:
            hbuf = torch.empty(min(cap, _PINNED_START), dtype=torch.uint8).pin_memory()
            keep["blob"] = hbuf
        st = _lib.load().urhgpu_outputs_to_host(pipe.ctx.handle, C.byref(o2), 1 if has_pos else 0, C.c_void_p(hbuf.data_ptr()), int(hbuf.numel()), C.byref(total))
        if st == _lib.ERR_CAPACITY and int(total.value) > hbuf.numel():
            # the blob is larger than the buffer so far: grow to what it needs (+ 25 %: the next, similar pass fits at once) and fetch again
            hbuf = torch.empty(min(cap, int(total.value) + int(total.value) // 4 + 4096), dtype=torch.uint8).pin_memory()
            keep["blob"] = hbuf
            st = _lib.load().urhgpu_outputs_to_host(pipe.ctx.handle, C.byref(o2), 1 if has_pos else 0, C.c_void_p(hbuf.data_ptr()), int(hbuf.numel()), C.byref(total))
        _lib.check(st)
        h = HostBits.from_blob(hbuf.data_ptr(), self.params, n_samples=int(self.qad.shape[0]) if self.qad is not None else 0,
                               d_qad=self.qad.data_ptr() if self.qad is not None else 0)
        h._keep = hbuf
        keep["owner"] = h
        if pool is None:
            self._hostbits = h
        return h

    def host_counts(self):
        """(n_rows, n_msg, n_bits, n_pos): one 32-byte D2H copy (synchronises)."""
        if self._host_counts is None:
            if self._ctx is not None:
                self._ctx.join()          # pipelined mode: the current stream waits for the tail of the pass
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

    def _through_blob(self):
        """Do the host accessors go through the compact blob?  Not for a result the pipeline did not make, and not when the pack kernel found
        a value its narrow types cannot hold (truncated bit 2: a capture of 2^31 samples and more -- or the NEGATIVE length / position the
        reference produces for a capture shorter than the tolerance, signal_functions.pyx:485-493): then the wide device tables are copied."""
        if self._pipe is None or self._outputs is None:
            return False
        return not (self.host().truncated & 4)

    def ppseq(self) -> np.ndarray:
        if self._through_blob():
            return self.host().check().ppseq()
        self.check_capacity()             # an overflowing table is clamped to cap_rows on the device: never hand that out
        n_rows = self.host_counts()[0]
        return self.rows_buf[:n_rows].cpu().numpy()

    def flat(self):
        """(bits u8, msg_off i64, pauses i64, pos i64, pos_off i64) on the host (fresh arrays: they outlive the pipeline's buffers)."""
        if self._through_blob():
            h = self.host().check()
            if self.pos_buf is None:
                return h.bits(), h.msg_off.copy(), h.pauses.copy(), np.zeros(0, np.int64), h.pos_off.copy()
            return h.flat()
        self.check_capacity()
        _, n_msg, n_bits, n_pos = self.host_counts()
        bits = self.bits_buf[:n_bits].cpu().numpy()
        msg_off = self.msg_off_buf[:n_msg + 1].cpu().numpy()
        pauses = self.pauses_buf[:n_msg].cpu().numpy()
        pos_off = self.pos_off_buf[:n_msg + 1].cpu().numpy()
        pos = self.pos_buf[:n_pos].cpu().numpy() if self.pos_buf is not None else np.zeros(0, np.int64)
        return bits, msg_off, pauses, pos, pos_off

    def to_host_pinned(self, pool: dict):
        """(ppseq, bits, msg_off, pauses, pos, pos_off) as numpy views of PINNED host buffers kept in `pool` (a dict owned by the
        caller: the views are valid until the next call with the same pool): one 40-byte read-back for the counts, then five
        asynchronous copies at PCIe speed and one synchronisation -- pageable `.cpu()` copies of the same 23 MB take three times as
        long (the timing window includes the D2H of the compact outputs)."""
        import torch
        self.check_capacity()
        n_rows, n_msg, n_bits, n_pos = self.host_counts()
        want = (("rows", self.rows_buf, n_rows), ("bits", self.bits_buf, n_bits), ("msg_off", self.msg_off_buf, n_msg + 1),
                ("pauses", self.pauses_buf, n_msg), ("pos", self.pos_buf, n_pos if self.pos_buf is not None else 0),
                ("pos_off", self.pos_off_buf, n_msg + 1))
        out = []
        for name, src, count in want:
            if src is None or count == 0:
                out.append(np.zeros((0, 2) if name == "rows" else 0, np.uint8 if name == "bits" else np.int64))
                continue
            shape = (count, 2) if name == "rows" else (count,)
            need = int(np.prod(shape))
            buf = pool.get(name)
            if buf is None or buf.numel() < need or buf.dtype != src.dtype:
                buf = torch.empty(max(need, 1024), dtype=src.dtype, pin_memory=True)
                pool[name] = buf
            dst = buf[:need].view(*shape)
            dst.copy_(src[:count], non_blocking=True)
            out.append(dst.numpy())
        torch.cuda.current_stream(self.rows_buf.device).synchronize()
        return tuple(out)

    def messages(self):
        """Reference-shaped result of _ppseq_to_bits: (list of array('B'), array('L'), list of array('L'))."""
        bits, off, pauses, pos, poff = self.flat()
        data = [array.array("B", bits[off[i]:off[i + 1]].tobytes()) for i in range(len(pauses))]
        pa = array.array("L", pauses.tolist())
        bsp = [array.array("L", pos[poff[i]:poff[i + 1]].tolist()) for i in range(len(pauses))] \
            if self.pos_buf is not None else []
        return data, pa, bsp

    def plain_bits_str(self):
        data, _, _ = self.messages()
        return ["".join(map(str, d)) for d in data]


class LazyDigitized:
    """(ppseq, bits, msg_off, pauses, pos, pos_off) of one digitisation as a sequence whose members are widened to the reference's
    types when somebody looks: what crossed PCIe is the compact blob (a HostBits over pinned memory), and a caller that only wants
    the bits never pays for the int64 pulse table.  materialize() widens everything and lets go of the pinned buffer."""

    def __init__(self, host: "HostBits", want_pos: bool = True):
        self._h = host.check()
        self._want_pos = want_pos
        self._v = [None] * 6

    def _get(self, i):
        if self._v[i] is None:
            h = self._h
            if i == 0:
                self._v[0] = h.ppseq()
            elif i == 1:
                self._v[1] = h.bits()
            elif i == 2:
                self._v[2] = h.msg_off.copy()
            elif i == 3:
                self._v[3] = h.pauses.copy()
            elif i == 4:
                self._v[4] = h.bit_sample_pos() if self._want_pos else np.zeros(0, np.int64)
            else:
                self._v[5] = h.pos_offsets() if self._want_pos else h.pos_off.copy()
        return self._v[i]

    def materialize(self):
        if self._h is not None:
            for i in range(6):
                self._get(i)
            self._h = None
        return self

    def __len__(self):
        return 6

    def __getitem__(self, k):
        if isinstance(k, slice):
            return tuple(self._get(i) for i in range(*k.indices(6)))
        return self._get(range(6)[k])

    def __iter__(self):
        return (self._get(i) for i in range(6))


class HostBits:
    """One pass's results ON THE HOST, from the compact blob (include/urhgpu.h "compact result blob"): numpy views of pinned memory
    owned by the stream -- valid until three pushes later; copy what has to live longer.  The accessors widen to the reference's own
    shapes and types: ppseq() is grab_pulse_lens' int64 (P, 2), bits() one byte per bit, bit_sample_pos() int64."""

    def __init__(self, r: "_lib.HostResult", params):
        self.seq, self.n_samples = int(r.seq), int(r.n_samples)
        self.n_rows, self.n_msg, self.n_bits, self.n_pos = int(r.n_rows), int(r.n_msg), int(r.n_bits), int(r.n_pos)
        self.rows_needed, self.truncated, self.blob_bytes = int(r.rows_needed), int(r.truncated), int(r.blob_bytes)
        self.params = params
        self.d_qad_ptr = r.d_qad
        # the views are made when somebody looks (a push per 0.3 ms must not allocate a dozen objects each)
        self._len16 = (r.row_len16, r.esc, int(r.n_esc)) if r.row_len16 else None      # 16-bit lengths + escape list (URHGPU_BLOB_LEN16)
        self._row16 = (r.row16, r.esc, int(r.n_esc)) if getattr(r, "row16", None) else None      # state | length words + escape list (URHGPU_BLOB_ROW16)
        self._ptr = dict(row_len=(r.row_len, self.n_rows, np.int32), row_state=(r.row_state, self.n_rows, np.int8),
                         bits_packed=(r.bits_packed, (self.n_bits + 7) // 8, np.uint8), msg_off=(r.msg_off, self.n_msg + 1, np.int64),
                         pauses=(r.pauses, self.n_msg, np.int64), pos_off=(r.pos_off, self.n_msg + 1, np.int64),
                         pos32=(r.pos32, self.n_pos, np.uint32) if r.pos32 else None)

    def __getattr__(self, name):
        spec = self.__dict__.get("_ptr", {}).get(name, False)
        if spec is False:
            raise AttributeError(name)
        if name in ("row_len", "row_state") and self.__dict__.get("_row16") is not None:
            # one uint16 per row: (state + 1) << 13 | length; a length field of 0x1FFF = look the row up in the escape list
            p16, pesc, n_esc = self._row16
            if self.n_rows <= 0:
                state, value = np.zeros(0, np.int8), np.zeros(0, np.int32)
            else:
                w = np.frombuffer((C.c_ubyte * (2 * self.n_rows)).from_address(p16), dtype=np.uint16, count=self.n_rows)
                state = ((w >> 13).astype(np.int16) - 1).astype(np.int8)
                value = (w & 0x1FFF).astype(np.int32)
                marked = value == 0x1FFF
                rows = np.zeros(0, np.int64)
                if n_esc > 0:
                    e = np.frombuffer((C.c_ubyte * (8 * n_esc)).from_address(pesc), dtype=np.uint32, count=2 * n_esc).reshape(-1, 2)
                    rows = e[:, 0].astype(np.int64)
                if int(marked.sum()) != n_esc or (n_esc and (rows.max() >= self.n_rows or not marked[rows].all() or len(np.unique(rows)) != n_esc)):
                    raise _lib.UrhGpuError(_lib.ERR_UNSUPPORTED, "compact blob: the packed rows and their escape list do not match")
                if n_esc > 0:
                    value[rows] = e[:, 1].copy().view(np.int32)
            self.__dict__["row_len"], self.__dict__["row_state"] = value, state
            return self.__dict__[name]
        if name == "row_len" and self.__dict__.get("_len16") is not None:
            # widen the shipped uint16 lengths; 0xFFFF = look the row up in the escape list ({uint32 row, int32 length} pairs)
            p16, pesc, n_esc = self._len16
            if self.n_rows <= 0:
                value = np.zeros(0, np.int32)
            else:
                value = np.frombuffer((C.c_ubyte * (2 * self.n_rows)).from_address(p16), dtype=np.uint16, count=self.n_rows).astype(np.int32)
                marked = value == 0xFFFF
                rows = np.zeros(0, np.int64)
                if n_esc > 0:
                    e = np.frombuffer((C.c_ubyte * (8 * n_esc)).from_address(pesc), dtype=np.uint32, count=2 * n_esc).reshape(-1, 2)
                    rows = e[:, 0].astype(np.int64)
                # every 0xFFFF has its entry and every entry its 0xFFFF (a true length of 65535 is an entry too)
                if int(marked.sum()) != n_esc or (n_esc and (rows.max() >= self.n_rows or not marked[rows].all() or len(np.unique(rows)) != n_esc)):
                    raise _lib.UrhGpuError(_lib.ERR_UNSUPPORTED, "compact blob: the 16-bit row lengths and their escape list do not match")
                if n_esc > 0:
                    value[rows] = e[:, 1].copy().view(np.int32)
            self.__dict__[name] = value
            return value
        if spec is None:
            value = None
        else:
            ptr, count, dtype = spec
            if not ptr or count <= 0:
                value = np.zeros(0, dtype)
            else:
                nbytes = count * np.dtype(dtype).itemsize
                value = np.frombuffer((C.c_ubyte * nbytes).from_address(ptr), dtype=dtype, count=count)
        self.__dict__[name] = value
        return value

    @classmethod
    def from_blob(cls, host_ptr: int, params, seq: int = 0, n_samples: int = 0, d_qad: int = 0):
        """the same object from a compact blob that lies in host memory at host_ptr (header first; include/urhgpu.h): what
        urhgpu_stream_* hands out for a single-GPU stream, made here for blobs copied by other means (sharded passes)"""
        hdr = np.frombuffer((C.c_ubyte * 128).from_address(host_ptr), dtype=np.int64, count=16)
        if int(hdr[0]) != _lib.BLOB_MAGIC:
            raise ValueError("not a result blob")
        r = _lib.HostResult()
        r.seq, r.n_samples = seq, n_samples
        r.n_rows, r.n_msg, r.n_bits, r.n_pos, r.rows_needed = (int(hdr[k]) for k in (1, 2, 3, 4, 5))
        r.blob_bytes = abs(int(hdr[6]))
        r.truncated = int(hdr[15])
        r.pauses, r.msg_off, r.pos_off = host_ptr + int(hdr[8]), host_ptr + int(hdr[9]), host_ptr + int(hdr[10])
        r.row_state, r.bits_packed, r.row_len = host_ptr + int(hdr[11]), host_ptr + int(hdr[12]), host_ptr + int(hdr[13])
        if int(hdr[7]) & _lib.BLOB_ROW16:
            n_esc = int(np.frombuffer((C.c_ubyte * 8).from_address(host_ptr + int(hdr[11])), dtype=np.int64, count=1)[0])
            r.row16, r.row_len, r.row_state, r.esc, r.n_esc = r.row_len, None, None, host_ptr + int(hdr[11]) + 8, abs(n_esc)
        elif int(hdr[7]) & _lib.BLOB_LEN16:
            off_esc = (int(hdr[12]) + (int(hdr[3]) + 7) // 8 + 15) & ~15
            n_esc = int(np.frombuffer((C.c_ubyte * 8).from_address(host_ptr + off_esc), dtype=np.int64, count=1)[0])
            r.row_len16, r.row_len, r.esc, r.n_esc = r.row_len, None, host_ptr + off_esc + 8, abs(n_esc)
        r.pos32 = host_ptr + int(hdr[14]) if int(hdr[7]) & 1 else None
        r.blob = host_ptr
        r.d_qad = d_qad or None
        return cls(r, params)

    def check(self):
        if self.truncated & 4:
            raise _lib.UrhGpuError(_lib.ERR_UNSUPPORTED, "a row length, position or state does not fit the compact blob's narrow types "
                                                         "(take the wide device outputs)")
        if self.truncated:
            raise _lib.UrhGpuError(_lib.ERR_CAPACITY, f"output capacity too small: the pulse table needs {self.rows_needed} rows")
        return self

    def ppseq(self) -> np.ndarray:
        out = np.empty((self.n_rows, 2), np.int64)
        out[:, 0] = self.row_state
        out[:, 1] = self.row_len
        if self.n_rows and self.row_state[0] == -128:        # a sharded ASK piece whose first row was merged into the previous rank's last
            return out[1:]                                   # row (URHGPU_ROW_ABSORBED): not a row of this piece, as ShardResult.piece() has it
        return out

    def bits(self) -> np.ndarray:
        return np.unpackbits(self.bits_packed, count=self.n_bits) if self.n_bits else np.zeros(0, np.uint8)

    def bit_sample_pos(self) -> np.ndarray:
        """All messages' bit_sample_pos back to back (int64; message m: [pos_offsets()[m], pos_offsets()[m + 1])).  With positions shipped
        (want_pos) they are the device's; without, they are derived HERE from the pulse table that was shipped -- they are a function
        of it (ProtocolAnalyzer.py:346-401 computes them from ppseq too), so not shipping them costs no information, only host time
        when somebody asks."""
        if self.pos32 is not None:
            return self.pos32.astype(np.int64)
        return self._derived_positions()[0]

    def pos_offsets(self) -> np.ndarray:
        return self.pos_off.copy() if self.pos32 is not None else self._derived_positions()[1]

    def _derived_positions(self):
        if self.__dict__.get("sharded_piece"):
            raise ValueError("a rank's piece of a sharded capture does not start at sample 0 of the capture: its positions are not derivable "
                             "from its rows alone -- run the pass with write_bit_sample_pos=True to have them shipped")
        if getattr(self, "_derived", None) is None:
            self._derived = positions_from_rows(self.row_state, self.row_len, self.params)
        return self._derived

    def flat(self):
        """(bits u8, msg_off i64, pauses i64, pos i64, pos_off i64): what BitsResult.flat() gives"""
        return self.bits(), self.msg_off.copy(), self.pauses.copy(), self.bit_sample_pos(), self.pos_offsets()


def positions_from_rows(row_state, row_len, p):
    """bit_sample_pos of every message from the pulse table, as ProtocolAnalyzer._ppseq_to_bits builds them (:346-411), on whole arrays:
    (positions int64, back to back; offsets int64[n_msg + 1]).  A message's entries: one position per bit (total_samples + k *
    samples_per_bit), then [start, end] of the long pause that closed it -- or the capture's total_samples for the trailing message."""
    st = np.asarray(row_state, dtype=np.int64)
    ln = np.asarray(row_len, dtype=np.int64)
    n = len(st)
    sps, bps, pt = int(p.samples_per_symbol), int(p.bits_per_symbol), int(p.pause_threshold)
    spb = int(sps / bps)
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(1, np.int64)
    ts = np.concatenate([[0], np.cumsum(ln)[:-1]])              # total_samples before each row
    q = ln // sps
    nsym = q + (2 * (ln - q * sps) > sps)                         # int(x) + (fraction > 0.5): exact for these integers
    pause = st == -1
    long_pause = pause & (nsym > pt) & (pt != 0)
    nbits = np.where(long_pause, 0, nsym * bps)
    data = ~pause & (nsym > 0)
    if pause[0]:                                                  # "Starts with Pause": only seeds total_samples
        nbits[0] = 0
        long_pause[0] = False
    grp = np.cumsum(long_pause) - long_pause                      # group of each row (a long pause closes its own group)
    n_grp = int(grp[-1]) + 1
    has_data = np.bincount(grp[data], minlength=n_grp) > 0
    # "elif not there_was_data": a long pause after a group without data drops what was gathered; such a group is no message
    closed = np.zeros(n_grp, dtype=bool)
    closed[grp[long_pause]] = True
    close_row = np.full(n_grp, -1, dtype=np.int64)
    close_row[grp[long_pause]] = np.nonzero(long_pause)[0]
    kept = has_data
    row_kept = kept[grp] & (nbits > 0)
    cnt = nbits[row_kept]
    per_bit_base = np.repeat(ts[row_kept], cnt)
    first = np.cumsum(cnt) - cnt
    k = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(first, cnt)
    bit_pos = per_bit_base + k * spb
    bits_per_group = np.bincount(grp[row_kept], weights=cnt, minlength=n_grp).astype(np.int64)
    msgs = np.nonzero(kept)[0]
    total_end = int(ts[-1] + ln[-1])
    out, off = [], [0]
    bit_cursor = np.concatenate([[0], np.cumsum(bits_per_group[msgs])])
    for j, g in enumerate(msgs.tolist()):
        out.append(bit_pos[bit_cursor[j]:bit_cursor[j + 1]])
        if closed[g]:
            r = int(close_row[g])
            out.append(np.array([ts[r], ts[r] + ln[r]], dtype=np.int64))
        else:
            out.append(np.array([total_end], dtype=np.int64))
        off.append(off[-1] + int(bits_per_group[g]) + (2 if closed[g] else 1))
    pos = np.concatenate(out) if out else np.zeros(0, np.int64)
    return pos.astype(np.int64), np.asarray(off, dtype=np.int64)


class CaptureStream:
    """Capture after capture with the results on the host (urhgpu_stream_*): push() queues a pass and returns at once with the result
    of the pass pushed three calls earlier (or None); flush() waits for the rest.  The hot kernel of pass i, the tail of pass i - 1
    and the D2H copy of pass i - 2's compact blob overlap."""

    def __init__(self, pipe: "DevicePipeline", n_max: int, p: DemodParams, want_qad=True, want_pos=True, dtype=np.float32, cap_rows=0, latency=None):
        """latency: True -- ONE capture at a time, its results as early as possible (a pass that finds the pipeline idle runs its tail in
        segments beside the hot kernel); False -- capture after capture, highest throughput (staged passes: rows through a staging blob in HBM and the runtime's copy); None: leave the context's
        setting (urhgpu_ctx_set_tuning "stream_latency") as it is"""
        if latency is not None:
            pipe.ctx.set_tuning("stream_latency", 1 if latency else 0)
        self.pipe, self.params = pipe, p
        self._cp = p.to_c(dtype)
        h = C.c_void_p()
        pipe.ctx.set_stream(pipe.torch.cuda.current_stream(pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_stream_create(pipe.ctx.handle, int(n_max), C.byref(self._cp), 1 if want_qad else 0,
                                                    1 if want_pos else 0, int(cap_rows), C.byref(h)))
        self._h = h
        self._dtype = np.dtype(dtype)
        # the captures of the passes that may still be running (three are in flight at most): a caller's temporary -- a pinned host buffer
        # under the DMA of push_upload in particular, which no allocator knows to be in use -- lives until its pass has been handed out
        self._inflight = []

    def _hold(self, *tensors):
        self._inflight.append(tensors)
        del self._inflight[:-4]

    def push(self, iq):
        torch = self.pipe.torch
        if iq.dtype == torch.complex64:
            iq = torch.view_as_real(iq)
        if _torch_dtype(iq) != self._dtype or not iq.is_contiguous():
            raise ValueError("the stream was created for contiguous (N, 2) captures of " + str(self._dtype))
        r = _lib.HostResult()
        self.pipe.ctx.set_stream(torch.cuda.current_stream(self.pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_stream_push(self._h, C.c_void_p(iq.data_ptr()), int(iq.shape[0]), C.byref(r)))
        self._hold(iq)
        return HostBits(r, self.params) if r.seq >= 0 else None

    def push_upload(self, host_iq, dev_iq):
        """A capture that is still on the host (urhgpu_stream_push_upload): host_iq -- a torch CPU tensor (pinned: PCIe speed) or a numpy
        array, (N, 2) of the stream's dtype or complex64 -- is copied into dev_iq (device tensor of the same shape: the capture stays
        resident there, e.g. as a Signal's data) piece by piece, and every piece is demodulated as it lands.  Returns like push()."""
        torch = self.pipe.torch
        if isinstance(host_iq, np.ndarray):
            host_iq = torch.from_numpy(host_iq)
        if host_iq.dtype == torch.complex64:
            host_iq = torch.view_as_real(host_iq)
        if dev_iq.dtype == torch.complex64:
            dev_iq = torch.view_as_real(dev_iq)
        if host_iq.is_cuda or not dev_iq.is_cuda or _torch_dtype(host_iq) != self._dtype or _torch_dtype(dev_iq) != self._dtype \
                or not host_iq.is_contiguous() or not dev_iq.is_contiguous() or host_iq.shape != dev_iq.shape:
            raise ValueError("push_upload: a contiguous host capture and a device tensor of the same shape, both of " + str(self._dtype))
        r = _lib.HostResult()
        self.pipe.ctx.set_stream(torch.cuda.current_stream(self.pipe.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_stream_push_upload(self._h, C.c_void_p(host_iq.data_ptr()), C.c_void_p(dev_iq.data_ptr()), int(host_iq.shape[0]),
                                                         C.byref(r)))
        self._hold(host_iq, dev_iq)
        return HostBits(r, self.params) if r.seq >= 0 else None

    def flush(self):
        arr = (_lib.HostResult * 3)()
        n = C.c_int(0)
        _lib.check(_lib.load().urhgpu_stream_flush(self._h, arr, C.byref(n)))
        self._inflight = []                                   # (every pass has finished)
        return [HostBits(arr[k], self.params) for k in range(n.value)]

    def stats(self) -> dict:
        out = (C.c_int64 * 4)()
        _lib.check(_lib.load().urhgpu_stream_stats(self._h, out))
        wide = C.c_int64(0)
        _lib.check(_lib.load().urhgpu_stream_wide_passes(self._h, C.byref(wide)))
        return {"pushed": int(out[0]), "short_copies": int(out[1]), "predicted_bytes": int(out[2]), "blob_capacity": int(out[3]),
                "wide_passes": int(wide.value)}

    def close(self):
        if self._h:
            _lib.load().urhgpu_stream_destroy(self._h)
            self._h = None
            self._inflight = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePipeline:
    """Owns a liburhgpu context bound to torch's current stream and the output buffers."""

    def __init__(self, device=None, pipelined=False, tuning=None, tail_stream_priority=0):
        """pipelined: back-to-back iq_to_bits passes overlap (the hot kernel of a pass runs while the tail of the previous
        one finishes on a second stream, see urhgpu_ctx_set_pipelined); results synchronise when they are read.
        tuning: {key: value} for urhgpu_ctx_set_tuning (A/B tooling); tail_stream_priority: torch stream priority of the tail's stream."""
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.UrhGpuError(_lib.ERR_NO_DEVICE, "no GPU visible to torch: the IQ->bits path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.ctx = _lib.Context(self.device.index)
        _lib.host_libm_verdict()                             # (once per process: a host libm the device code does not restate is reported)
        self._bufs = {}
        self._pinned = {}                                   # pinned host buffers of BitsResult.host()
        self.tail_stream = None
        for key, value in (tuning or {}).items():
            self.ctx.set_tuning(key, value)
        if pipelined:
            self.tail_stream = torch.cuda.Stream(self.device, priority=int(tail_stream_priority))
            self.ctx.set_pipelined(True, self.tail_stream.cuda_stream)

    def tail_context(self):
        """torch stream context of the work that follows the hot kernel (pipelined mode), else a no-op context"""
        import contextlib
        return self.torch.cuda.stream(self.tail_stream) if self.tail_stream is not None else contextlib.nullcontext()

    def halo_context(self):
        """tail_context for work that reads what the caller's stream has produced: the tail stream waits for it first"""
        if self.tail_stream is not None:
            self.tail_stream.wait_stream(self.torch.cuda.current_stream(self.device))
        return self.tail_context()

    def join(self):
        self.ctx.join()

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        need = int(np.prod(shape))
        if t is None or t.numel() < need or t.dtype != dtype:
            t = self.torch.empty(need, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:need].view(*shape)

    def capacities(self, n: int, p: DemodParams, cap_rows=None):
        sps = max(int(p.samples_per_symbol), 1)
        if cap_rows is None:
            # accepted runs cannot be denser than one per (tolerance+1) samples; the default assumes
            # at most ~4 per symbol and is checked (ERR_CAPACITY) after the fact
            cap_rows = min(n // (p.tolerance + 1) + 2, max(4096, 4 * (n // sps) + 4096))
        cap_bits = (n // sps + 2 * cap_rows // 8 + 64) * int(p.bits_per_symbol) + cap_rows
        cap_msg = max(64, cap_rows // 4)
        cap_pos = cap_bits + 2 * cap_msg + 2
        return cap_rows, cap_bits, cap_msg, cap_pos

    def reserve(self, n: int, p: DemodParams):
        self.ctx.reserve(n, p.tolerance)

    def stream(self, n_max: int, p: DemodParams, want_qad=True, want_pos=True, dtype=np.float32, cap_rows=0, latency=None) -> CaptureStream:
        """a CaptureStream on this pipeline's context (which it switches to pipelined passes)"""
        return CaptureStream(self, n_max, p, want_qad, want_pos, dtype, cap_rows, latency)

    def iq_to_bits(self, iq, p: DemodParams, want_qad=True, cap_rows=None, slot=0) -> BitsResult:
        """iq: torch tensor on this device, shape (N, 2) of int8/uint8/int16/uint16/float32, or complex64 (N,).
        The result lives in buffers owned by the pipeline and is overwritten by the next pass with the same `slot`."""
        torch = self.torch
        if iq.dtype == torch.complex64:
            iq = torch.view_as_real(iq)
        if iq.dim() != 2 or iq.shape[1] != 2 or not iq.is_contiguous():
            raise ValueError("IQ must be a contiguous (N, 2) tensor")
        npdt = _torch_dtype(iq)
        n = iq.shape[0]
        cp = p.to_c(npdt)
        cap_rows, cap_bits, cap_msg, cap_pos = self.capacities(n, p, cap_rows)
        sfx = f"s{slot}:" if slot else ""
        qad = self._buf(sfx + "qad", (n,), torch.float32) if want_qad else None
        rows = self._buf(sfx + "rows", (cap_rows, 2), torch.int64)
        bits = self._buf(sfx + "bits", (cap_bits,), torch.uint8)
        msg_off = self._buf(sfx + "msg_off", (cap_msg + 1,), torch.int64)
        pauses = self._buf(sfx + "pauses", (cap_msg,), torch.int64)
        pos_off = self._buf(sfx + "pos_off", (cap_msg + 1,), torch.int64)
        pos = self._buf(sfx + "pos", (cap_pos,), torch.int64) if p.write_bit_sample_pos else None
        counts = self._buf(sfx + "counts", (5,), torch.int64)
        o = _lib.Outputs()
        o.qad = qad.data_ptr() if qad is not None else None
        o.rows = rows.data_ptr(); o.cap_rows = cap_rows
        o.bits = bits.data_ptr(); o.cap_bits = cap_bits
        o.msg_off = msg_off.data_ptr(); o.pauses = pauses.data_ptr(); o.cap_msg = cap_msg
        o.pos = pos.data_ptr() if pos is not None else None
        o.cap_pos = cap_pos if pos is not None else 0
        o.pos_off = pos_off.data_ptr()
        o.counts = counts.data_ptr()
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_iq_to_bits_dev(self.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp), C.byref(o)))
        return BitsResult(qad, rows, bits, msg_off, pauses, pos, pos_off, counts, p, self.ctx, pipe=self, outputs=o)

    def iq_to_bits_checked(self, iq, p: DemodParams, want_qad=True) -> BitsResult:
        """iq_to_bits with the output capacities verified (one 40-byte read-back) and, if the default capacities were
        too small -- a capture that is mostly noise produces far more pulse-table rows than 4 per symbol --, one more
        pass with what the first one reported."""
        res = self.iq_to_bits(iq, p, want_qad)
        n_rows, n_msg, n_bits, n_pos = res.host_counts()
        need = res._rows_needed
        if need > res.rows_buf.shape[0] or n_msg > res.pauses_buf.shape[0] or n_bits > res.bits_buf.shape[0] \
                or (res.pos_buf is not None and n_pos > res.pos_buf.shape[0]):
            n = iq.shape[0]
            worst = n // (p.tolerance + 1) + 2
            res = self.iq_to_bits(iq, p, want_qad, cap_rows=min(worst, max(2 * need, 4096)))
            res.check_capacity()
        return res

    def qad_to_bits(self, qad, p: DemodParams, cap_rows=None, slot=0) -> BitsResult:
        """grab_pulse_lens + _ppseq_to_bits on an already demodulated signal (float32 (N,) on this device) -- what the reference does
        whenever only slicing parameters changed, and after AutoInterpretation.estimate has demodulated the capture (Signal.qad is
        cached, Signal.py:421-431): 4 B per sample read instead of the IQ stream again.  Outputs as iq_to_bits (qad = the input)."""
        torch = self.torch
        if qad.dtype != torch.float32 or qad.dim() != 1 or not qad.is_contiguous():
            raise ValueError("Buffer dtype mismatch, expected 'float' (grab_pulse_lens takes float[::1])")
        n = int(qad.shape[0])
        cp = p.to_c(np.float32)
        if cap_rows is None:
            cap_rows = n // (p.tolerance + 1) + 2            # exact bound: one pass, no retry (the table is sliced from a resident signal)
        cap_rows, cap_bits, cap_msg, cap_pos = self.capacities(n, p, cap_rows)
        sfx = f"q{slot}:"
        rows = self._buf(sfx + "rows", (cap_rows, 2), torch.int64)
        n_rows = self._buf(sfx + "n_rows", (1,), torch.int64)
        bits = self._buf(sfx + "bits", (cap_bits,), torch.uint8)
        msg_off = self._buf(sfx + "msg_off", (cap_msg + 1,), torch.int64)
        pauses = self._buf(sfx + "pauses", (cap_msg,), torch.int64)
        pos_off = self._buf(sfx + "pos_off", (cap_msg + 1,), torch.int64)
        pos = self._buf(sfx + "pos", (cap_pos,), torch.int64) if p.write_bit_sample_pos else None
        counts = self._buf(sfx + "counts", (5,), torch.int64)
        lib = _lib.load()
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        if n == 0:
            counts.zero_()
            msg_off[:1] = 0
            pos_off[:1] = 0
            return BitsResult(qad, rows, bits, msg_off, pauses, pos, pos_off, counts, p, None)
        _lib.check(lib.urhgpu_grab_pulse_lens_dev(self.ctx.handle, C.c_void_p(qad.data_ptr()), n, C.byref(cp), C.c_void_p(rows.data_ptr()), cap_rows,
                                                  C.c_void_p(n_rows.data_ptr())))
        o = _lib.Outputs()
        o.qad = None
        o.rows = rows.data_ptr(); o.cap_rows = cap_rows
        o.bits = bits.data_ptr(); o.cap_bits = cap_bits
        o.msg_off = msg_off.data_ptr(); o.pauses = pauses.data_ptr(); o.cap_msg = cap_msg
        o.pos = pos.data_ptr() if pos is not None else None
        o.cap_pos = cap_pos if pos is not None else 0
        o.pos_off = pos_off.data_ptr(); o.counts = counts.data_ptr()
        _lib.check(lib.urhgpu_ppseq_to_bits_dev(self.ctx.handle, C.c_void_p(rows.data_ptr()), C.c_void_p(n_rows.data_ptr()), cap_rows, C.byref(cp),
                                                C.byref(o)))
        return BitsResult(qad, rows, bits, msg_off, pauses, pos, pos_off, counts, p, None, pipe=self, outputs=o)

    def afp_demod(self, iq, p: DemodParams):
        torch = self.torch
        if iq.dtype == torch.complex64:
            iq = torch.view_as_real(iq)
        npdt = _torch_dtype(iq)
        n = iq.shape[0]
        qad = torch.empty(n, dtype=torch.float32, device=self.device)
        cp = p.to_c(npdt, demod_only=True)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().urhgpu_afp_demod_dev(self.ctx.handle, C.c_void_p(iq.data_ptr()), n, C.byref(cp),
                                                    C.c_void_p(qad.data_ptr())))
        return qad
