"""Executable model (pure Python, small inputs) of the SHARDED run-segmentation / bit-expansion path:
what each rank's HIP engine (urh_amd/shard_engine.py -> urhgpu_shard_* in liburhgpu.so) computes between
the all-gathers of urh_amd/sharding.py.  It plugs into the same orchestration (ShardedPipeline), so the CPU
suite can run the real multi-process protocol over gloo and compare the stitched result with the oracle.
TEST INFRASTRUCTURE ONLY.

The model's shard input is the demodulated signal (float32 qad) rather than IQ: demodulation itself has
no cross-shard state beyond the 2-sample halo, which the GPU tests cover."""
import numpy as np
import torch

import model_runs as m

NONE = m.NONE
ABSORBED = -(1 << 62)       # state value of an ASK head row that merged into the previous rank's last row


def resolve_table(chunks, tol, init_state, local_pass):
    """pulse_table.hip resolve stage over a chunk table.  local_pass: the table is ONE shard on its own --
    the state before it is unknown (NONE: its first stable run counts as accepted, tentatively) and a
    trailing short run that reaches the shard end stays open (reported, not decided)."""
    nc = len(chunks)
    open_chunk = -1
    for c, ch in enumerate(chunks):
        ps = False
        if ch["pend_pos"] >= 0:
            ln = ch["start"] + ch["len"] - ch["pend_pos"]
            u, hit = c + 1, False
            while ln <= tol and u < nc:
                ln += chunks[u]["lead"]
                if chunks[u]["lead"] < chunks[u]["len"]:
                    hit = True
                    break
                u += 1
            ps = ln > tol
            if local_pass and not ps and not hit:
                open_chunk = c
        ch["pend_stable"] = ps
    prev_state = NONE if local_pass else init_state
    prev_acc = (-1, init_state)
    total = 0
    first_stable = last_stable = last_acc_chunk = -1
    for c, ch in enumerate(chunks):
        cnt, ps = ch["cnt"], ch["pend_stable"]
        first_acc = cnt > 0 and ch["first_state"] != prev_state
        before_pend = ch["last_state"] if cnt > 0 else prev_state
        pend_acc = ps and ch["pend_state"] != before_pend
        ch["first_acc"], ch["pend_acc"] = first_acc, pend_acc
        ch["out_off"] = total
        ch["prev_pos"], ch["prev_state"] = prev_acc
        total += (cnt - 1 + first_acc if cnt > 0 else 0) + pend_acc
        if pend_acc:
            prev_acc = (ch["pend_pos"], ch["pend_state"])
        if pend_acc or cnt >= 2 or (cnt == 1 and first_acc):
            if not pend_acc:
                prev_acc = (ch["last_pos"], ch["last_state"])
            last_acc_chunk = c
        if ps or cnt > 0:
            if first_stable < 0:
                first_stable = c
            last_stable = c
            prev_state = ch["pend_state"] if ps else ch["last_state"]
    return dict(total=total, last_acc=prev_acc, open_chunk=open_chunk, first_stable=first_stable,
                last_stable=last_stable, last_acc_chunk=last_acc_chunk)


def shard_summary(chunks, tol, init_state):
    """One ChunkInfo that stands for the whole shard in the other ranks' tables."""
    r = resolve_table(chunks, tol, init_state, local_pass=True)
    first_nonlead = next((c for c, ch in enumerate(chunks) if ch["lead"] < ch["len"]), -1)
    total_len = sum(ch["len"] for ch in chunks)
    s = dict(pend_pos=-1, pend_state=0, start=chunks[0]["start"], len=total_len, init_state=init_state)
    s["lead"] = total_len if first_nonlead < 0 else \
        chunks[first_nonlead]["start"] - chunks[0]["start"] + chunks[first_nonlead]["lead"]
    if r["open_chunk"] >= 0:
        s["pend_pos"], s["pend_state"] = chunks[r["open_chunk"]]["pend_pos"], chunks[r["open_chunk"]]["pend_state"]
    s["cnt"] = r["total"]
    f, l = r["first_stable"], r["last_stable"]
    s["first_state"] = NONE if f < 0 else (chunks[f]["first_state"] if chunks[f]["cnt"] > 0 else chunks[f]["pend_state"])
    s["last_state"] = NONE if l < 0 else (chunks[l]["pend_state"] if chunks[l]["pend_stable"] else chunks[l]["last_state"])
    s["last_pos"] = r["last_acc"][0] if r["last_acc_chunk"] >= 0 else 0
    return s


_FIELDS = ["pend_pos", "lead", "start", "len", "last_pos", "cnt", "first_state", "last_state", "pend_state", "init_state"]


def pack(s):
    return torch.tensor([int(s[k]) for k in _FIELDS], dtype=torch.int64)


def unpack(t):
    return {k: int(v) for k, v in zip(_FIELDS, t.tolist())}


def nsym(ln, sps):
    f = ln / sps
    k = int(f)
    if f - k > 0.5:
        k += 1
    return k


class ModelShardEngine:
    ctx = None

    def __init__(self, tile=32, span=8, chunk_tiles=2):
        self.tile, self.span, self.chunk_tiles = tile, span, chunk_tiles

    def reserve(self, n_local, p):
        pass

    # -- phase 0: halo ------------------------------------------------------------------------------------
    def tail(self, qad_local, p):
        return torch.tensor([float(qad_local[-1])], dtype=torch.float32)

    # -- phase 1: local run segmentation + shard summary ---------------------------------------------------
    def runs(self, qad_local, left, pos_base, n_total, rank, world, p, want_qad):
        from urh_amd.signal_functions import noise_for_mod_type
        self.p, self.rank, self.world, self.n_total = p, rank, world, n_total
        self.noise_val = np.float32(noise_for_mod_type(p.modulation_type))
        order = 2 ** p.bits_per_symbol
        n2 = order // 2
        c, sp = np.float32(p.center), np.float32(p.center_spacing)
        thr = [c - np.float32(n2 - (i + 1)) * sp for i in range(n2)] + [c + np.float32(i + 1 - n2) * sp for i in range(n2, order - 1)]
        self.thr = thr
        q = np.asarray(qad_local, dtype=np.float32)
        states = [m.classify(float(x), self.noise_val, thr) for x in q]
        prev8 = m.ST_NONE if rank == 0 else m.classify(float(left[0]), self.noise_val, thr)
        init_state = 0
        if rank == 0:
            init_state = 0 if q[0] == self.noise_val else m.classify(0.0, self.noise_val, thr, check_noise=False)
        chunk_len = self.tile * self.chunk_tiles
        self.chunks, self.slabs = [], []
        for a0 in range(0, len(q), chunk_len):
            a1 = min(a0 + chunk_len, len(q))
            info, slab = m.chunk_pass(states, a0, a1, prev8 if a0 == 0 else states[a0 - 1], p.tolerance, self.tile, self.span)
            info["start"] += pos_base
            info["last_pos"] += pos_base
            if info["pend_pos"] >= 0:
                info["pend_pos"] += pos_base
            self.chunks.append(info)
            self.slabs.append([(pos + pos_base, st) for pos, st in slab])
        return pack(shard_summary(self.chunks, p.tolerance, init_state))

    # -- phase 2: global resolve on [summaries before | local chunks | summaries after], local rows ------------
    def rows(self, summaries):
        p, r, W = self.p, self.rank, self.world
        S = [unpack(summaries[k]) for k in range(W)]
        for s in S:
            s["cnt"] = int(s["cnt"])
        table = S[:r] + self.chunks + S[r + 1:]
        res = resolve_table(table, p.tolerance, table[0]["init_state"] if r > 0 else S[0]["init_state"], local_pass=False)
        P, last_acc = res["total"], res["last_acc"]
        row_base = self.chunks[0]["out_off"]
        is_ask = p.modulation_type == "ASK"
        rows = []
        self.ts_carry = 0
        for ch, slab in zip(self.chunks, self.slabs):
            skip = 1 if (ch["cnt"] > 0 and not ch["first_acc"]) else 0
            recs = slab[skip:] + ([(ch["pend_pos"], ch["pend_state"])] if ch["pend_acc"] else [])
            for j, (pos, st) in enumerate(recs):
                ppos, pst = (ch["prev_pos"], ch["prev_state"]) if j == 0 else recs[j - 1]
                g = ch["out_off"] + j
                ln = pos + 1 if g == 0 else pos - ppos
                state = pst - 1
                if is_ask and state == -1 and ln < p.samples_per_symbol:
                    state = 0
                if g == row_base:
                    self.ts_carry = 0 if g == 0 else ppos + 1
                assert g - row_base == len(rows)
                rows.append([state, ln])
        if r == W - 1 and P < self.n_total:
            ln = (self.n_total - p.tolerance) if P == 0 else (self.n_total - 1 - last_acc[0] - p.tolerance)
            if not rows:
                self.ts_carry = 0 if P == 0 else last_acc[0] + 1
            rows.append([last_acc[1] - 1, ln])
        self.row_base = row_base
        if not is_ask:
            self.rows_local = rows
            return None
        merged = []
        for st, ln in rows:
            if merged and merged[-1][0] == st:
                merged[-1][1] += ln
            else:
                merged.append([st, ln])
        self.rows_local = merged
        if not merged:
            return torch.tensor([0, 0, 0, 0, 0], dtype=torch.int64)
        return torch.tensor([len(merged), merged[0][0], merged[0][1], merged[-1][0], merged[-1][1]], dtype=torch.int64)

    # -- phase 3: ASK cross-shard merge, per-row bit counts, boundary flags ------------------------------------
    def bits_prepare(self, merged_all):
        p, r, W = self.p, self.rank, self.world
        rows = self.rows_local
        if merged_all is not None and rows:
            M = merged_all.tolist()
            prev = next((q for q in range(r - 1, -1, -1) if M[q][0] > 0), -1)
            absorbed = prev >= 0 and M[prev][3] == rows[0][0]
            if not (absorbed and len(rows) == 1):
                last_state = rows[-1][0]
                for q in range(r + 1, W):
                    if M[q][0] == 0:
                        continue
                    if M[q][1] != last_state:
                        break
                    rows[-1][1] += M[q][2]
                    if M[q][0] > 1:
                        break
            self.absorbed_total = None
            if absorbed:
                if len(rows) == 1 and rows[0][0] == -1:
                    # my only row continues a pause owned by an earlier rank: its full length (the reference's
                    # last-row pause, ProtocolAnalyzer.py:411) is the owner's last row + everything absorbed into it
                    tot = rows[0][1]
                    for q in range(r + 1, W):
                        tot += M[q][2] if M[q][0] > 0 else 0
                    for q in range(r - 1, -1, -1):
                        if M[q][0] == 0:
                            continue
                        tot += M[q][4]
                        if M[q][0] > 1:
                            break
                        pq = next((u for u in range(q - 1, -1, -1) if M[u][0] > 0), -1)
                        if not (pq >= 0 and M[pq][3] == M[q][1]):
                            break
                    self.absorbed_total = tot
                rows[0][0] = ABSORBED
        sps, bps, pt = p.samples_per_symbol, p.bits_per_symbol, p.pause_threshold
        vals = []
        for i, (t, ln) in enumerate(rows):
            v = [0, 0, ln, 0]
            if t != ABSORBED and not (self.row_base == 0 and i == 0 and t == -1):
                ns = nsym(ln, sps)
                if t == -1:
                    if ns <= pt or pt == 0:
                        v[0] = ns * bps if ns > 0 else 0
                    else:
                        v[1] = 1
                else:
                    v[0] = ns * bps if ns > 0 else 0
                    v[3] = 1 if ns > 0 else 0
            vals.append(v)
        self.vals = vals
        n_l = sum(v[1] for v in vals)
        head_d = tail_d = 0
        seen_l = 0
        for v in vals:
            if v[1]:
                seen_l += 1
            elif v[3]:
                if seen_l == 0:
                    head_d = 1
                if seen_l == n_l:
                    tail_d = 1
        return torch.tensor([1 if n_l else 0, head_d, tail_d], dtype=torch.int64)

    # -- phase 4: groups -> messages, expansion --------------------------------------------------------------
    def bits_finish(self, flags_all):
        p, r, W = self.p, self.rank, self.world
        F = flags_all.tolist()
        head_extra = 0
        for q in range(r - 1, -1, -1):
            head_extra |= F[q][2]
            if F[q][0]:
                break
        tail_extra = 0
        for q in range(r + 1, W):
            tail_extra |= F[q][1]
            if F[q][0]:
                break
        rows, vals = self.rows_local, self.vals
        sps, bps = p.samples_per_symbol, p.bits_per_symbol
        spb = int(sps / bps)
        is_last = r == W - 1
        # local groups
        groups, cur = [], dict(rows=[], data=0, closed=0, pause=0)
        for i, v in enumerate(vals):
            if v[1]:
                cur["closed"], cur["pause"], cur["close_row"] = 1, v[2], i
                groups.append(cur)
                cur = dict(rows=[], data=0, closed=0, pause=0)
            else:
                cur["rows"].append(i)
                cur["data"] += v[3]
        if rows:
            cur["pause"] = rows[-1][1] if rows[-1][0] == -1 else 0
            if rows[-1][0] == ABSORBED and getattr(self, "absorbed_total", None) is not None:
                cur["pause"] = self.absorbed_total
            groups.append(cur)            # trailing (open unless this is the last rank)
        ts = [self.ts_carry]
        for v in vals:
            ts.append(ts[-1] + v[2])
        bits, pos, msg_end, pos_end, pauses = [], [], [], [], []
        for gi, g in enumerate(groups):
            is_msg = g["data"] > 0 or (gi == 0 and head_extra) or (gi == len(groups) - 1 and not g["closed"] and tail_extra)
            if not is_msg:
                continue
            for i in g["rows"]:
                t = rows[i][0]
                for k in range(vals[i][0]):
                    bits.append(0 if t < 0 else (t >> (bps - 1 - k % bps)) & 1)
                    pos.append(ts[i] + k * spb)
            if g["closed"]:
                tc = ts[g["close_row"]]
                pos.extend([tc, tc + g["pause"]])
            elif is_last:
                pos.append(ts[len(rows)])
            if g["closed"] or is_last:
                msg_end.append(len(bits))
                pos_end.append(len(pos))
                pauses.append(g["pause"])
        out_rows = [row for row in rows if row[0] != ABSORBED]
        return dict(rows=np.array(out_rows, dtype=np.int64).reshape(-1, 2), bits=np.array(bits, np.uint8),
                    msg_end=np.array(msg_end, np.int64), pauses=np.array(pauses, np.int64),
                    pos=np.array(pos, np.int64), pos_end=np.array(pos_end, np.int64))
