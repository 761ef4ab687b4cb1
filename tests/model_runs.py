"""Executable model (pure Python, small inputs) of the chunk/tile run-segmentation algorithm the HIP
kernels implement (urh_amd/csrc/demod_runs.hip, pulse_table.hip).  It exists so that the ALGORITHM
(tiles, carried short runs, tentative first records, chunk resolution, group logic of the bit
expansion) can be checked against the oracle on the CPU with tiny tile sizes that make every
boundary case frequent.  TEST INFRASTRUCTURE ONLY.
"""
import numpy as np

NONE = 0xFFFF
ST_NONE = 0xFF


def classify(q, noise_val, thr, check_noise=True):
    """state byte: 0 = pause, k+1 = state k (signal_functions.pyx:435-442)"""
    if check_noise and q == noise_val:
        return 0
    st = len(thr)
    for k, t in enumerate(thr):
        if q <= t:
            st = k
            break
    return st + 1


def chunk_pass(states, a0, a1, prev_state8, tol, tile, span):
    """k_demod_runs phase 2 for one chunk.  states: state bytes of the whole capture."""
    slab = []
    pend_pos, pend_state, lead, carry_last, first_state, last_pos = -1, 0, -1, NONE, NONE, 0
    for ta in range(a0, a1, tile):
        tv = min(tile, a1 - ta)
        st = states[ta:ta + tv]
        nthreads = (tile + span - 1) // span
        # boundary masks
        bm = [0] * nthreads
        for j in range(tv):
            prev = prev_state8 if j == 0 else st[j - 1]
            if st[j] != prev:
                bm[j // span] |= 1 << (j % span)
        bpos = [t * span + b for t in range(nthreads) for b in range(span) if bm[t] >> b & 1]
        first = bpos[0] if bpos else tile
        last = bpos[-1] if bpos else -1
        # stable masks (per thread, bounded look-ahead)
        stable = [0] * nthreads
        newpend = -1
        for t in range(nthreads):
            if not bm[t]:
                continue
            q = tv
            limit = span * t + (span - 1) + tol + 1
            u = t + 1
            while u < nthreads and span * u < limit and span * u < tv:
                if bm[u]:
                    q = span * u + (bm[u] & -bm[u]).bit_length() - 1
                    break
                u += 1
            m = bm[t]
            while m:
                hi = m.bit_length() - 1
                pos = span * t + hi
                if q - pos > tol:
                    stable[t] |= 1 << hi
                elif pos == last:
                    newpend = pos
                q = pos
                m &= ~(1 << hi)
        # thread 0: settle the carried run, lead
        if lead < 0 and first < tile:
            lead = (ta - a0) + first
        if pend_pos >= 0:
            end = ta + (first if first < tile else tv)
            decided = first < tile or end - pend_pos > tol
            if decided:
                if end - pend_pos > tol:
                    if pend_state != carry_last:
                        if not slab:
                            first_state = pend_state
                        slab.append((pend_pos, pend_state))
                        last_pos = pend_pos
                    carry_last = pend_state
                pend_pos = -1
        # dedupe + emit
        prev = carry_last
        for t in range(nthreads):
            m = stable[t]
            while m:
                lo = (m & -m).bit_length() - 1
                s = st[span * t + lo]
                if s != prev:
                    if not slab:
                        first_state = s
                    slab.append((ta + span * t + lo, s))
                    last_pos = ta + span * t + lo
                prev = s
                m &= m - 1
        carry_last = prev
        prev_state8 = st[tv - 1]
        if newpend >= 0:
            pend_pos, pend_state = ta + newpend, st[newpend]
    info = dict(pend_pos=pend_pos, pend_state=pend_state, lead=(a1 - a0) if lead < 0 else lead, start=a0, len=a1 - a0,
                cnt=len(slab), first_state=first_state, last_state=carry_last, last_pos=last_pos)
    return info, slab


def resolve(chunks, tol, init_state):
    """k_resolve_chunks"""
    nc = len(chunks)
    prev_state = init_state
    prev_acc = (-1, init_state)
    total = 0
    for c in range(nc):
        ch = chunks[c]
        ps = False
        if ch["pend_pos"] >= 0:
            ln = ch["start"] + ch["len"] - ch["pend_pos"]
            u = c + 1
            while ln <= tol and u < nc:
                ln += chunks[u]["lead"]
                if chunks[u]["lead"] < chunks[u]["len"]:
                    break
                u += 1
            ps = ln > tol
        cnt = ch["cnt"]
        first_acc = cnt > 0 and ch["first_state"] != prev_state
        before_pend = ch["last_state"] if cnt > 0 else prev_state
        pend_acc = ps and ch["pend_state"] != before_pend
        ch["first_acc"], ch["pend_acc"] = first_acc, pend_acc
        ch["out_off"] = total
        ch["prev_pos"], ch["prev_state"] = prev_acc
        total += (cnt - 1 + first_acc if cnt > 0 else 0) + pend_acc
        if pend_acc:
            prev_acc = (ch["pend_pos"], ch["pend_state"])
        elif cnt >= 2 or (cnt == 1 and first_acc):
            prev_acc = (ch["last_pos"], ch["last_state"])
        if ps:
            prev_state = ch["pend_state"]
        elif cnt > 0:
            prev_state = ch["last_state"]
    return total, prev_acc


def emit_rows(chunks, slabs, total, last_acc, n, tol, is_ask, sps):
    """k_emit_rows + final row (+ ASK merge)"""
    rows = [None] * total
    for ch, slab in zip(chunks, slabs):
        skip = 1 if (ch["cnt"] > 0 and not ch["first_acc"]) else 0
        recs = slab[skip:] + ([(ch["pend_pos"], ch["pend_state"])] if ch["pend_acc"] else [])
        for j, (pos, st) in enumerate(recs):
            ppos, pst = (ch["prev_pos"], ch["prev_state"]) if j == 0 else recs[j - 1]
            g = ch["out_off"] + j
            ln = pos + 1 if g == 0 else pos - ppos
            state = pst - 1
            if is_ask and state == -1 and ln < sps:
                state = 0
            rows[g] = [state, ln]
    if total < n:
        ln = (n - tol) if total == 0 else (n - 1 - last_acc[0] - tol)
        rows.append([last_acc[1] - 1, ln])
    if is_ask:
        merged = []
        for st, ln in rows:
            if merged and merged[-1][0] == st:
                merged[-1][1] += ln
            else:
                merged.append([st, ln])
        rows = merged
    return np.array(rows, dtype=np.int64).reshape(-1, 2)


def grab_pulse_lens_model(samples, center_thresholds, noise_val, tol, is_ask, sps, tile=64, span=8, chunk_tiles=2):
    n = len(samples)
    if n == 0:
        return np.zeros((0, 2), np.int64)
    states = [classify(float(q), noise_val, list(center_thresholds)) for q in samples]
    # :421-429 -- the literal 0.0 is classified by the thresholds only (no NOISE test)
    init_state = 0 if samples[0] == noise_val else classify(0.0, noise_val, list(center_thresholds), check_noise=False)
    chunk_len = tile * chunk_tiles
    chunks, slabs = [], []
    for a0 in range(0, n, chunk_len):
        a1 = min(a0 + chunk_len, n)
        prev8 = ST_NONE if a0 == 0 else states[a0 - 1]
        info, slab = chunk_pass(states, a0, a1, prev8, tol, tile, span)
        chunks.append(info)
        slabs.append(slab)
    total, last_acc = resolve(chunks, tol, init_state)
    return emit_rows(chunks, slabs, total, last_acc, n, tol, is_ask, sps)


# ---- model of the device _ppseq_to_bits (group logic) -----------------------------------------------
def ppseq_to_bits_model(rows, sps, bps, write_pos, pause_threshold):
    rows = np.asarray(rows, dtype=np.int64).reshape(-1, 2)
    n = len(rows)
    spb = int(sps / bps)

    def nsym(ln):
        f = ln / sps
        k = int(f)
        if f - k > 0.5:
            k += 1
        return k
    vals = []
    for i in range(n):
        t, ln = int(rows[i, 0]), int(rows[i, 1])
        v = [0, 0, ln, 0]
        if not (i == 0 and t == -1):
            ns = nsym(ln)
            if t == -1:
                if ns <= pause_threshold or pause_threshold == 0:
                    v[0] = ns * bps if ns > 0 else 0
                else:
                    v[1] = 1
            else:
                v[0] = ns * bps if ns > 0 else 0
                v[3] = 1 if ns > 0 else 0
        vals.append(v)
    ex = [[0, 0, 0, 0]]
    for v in vals:
        ex.append([ex[-1][k] + v[k] for k in range(4)])
    groups = {}
    for i, v in enumerate(vals):
        e = ex[i]
        if v[1]:
            groups[e[1]] = dict(bits_end=e[0], data_end=e[3], ts_close=e[2], pause=v[2], closed=1)
        if i + 1 == n:
            groups[e[1] + v[1]] = dict(bits_end=e[0] + v[0], data_end=e[3] + v[3], ts_close=e[2] + v[2],
                                       pause=int(rows[i, 1]) if rows[i, 0] == -1 else 0, closed=0)
    ng = (ex[-1][1] + 1) if n else 0
    bits, msg_off, pauses, pos, pos_off = [], [0], [], [], [0]
    gout = []
    for g in range(ng):
        gi = groups[g]
        d0 = groups[g - 1]["data_end"] if g else 0
        b0 = groups[g - 1]["bits_end"] if g else 0
        is_msg = gi["data_end"] - d0 > 0
        gout.append(dict(bits_start=b0, out_bits=len(bits), out_pos=len(pos), is_msg=is_msg))
        if is_msg:
            nb = gi["bits_end"] - b0
            bits.extend([None] * nb)
            if write_pos:
                pos.extend([None] * nb)
                pos.extend([gi["ts_close"], gi["ts_close"] + gi["pause"]] if gi["closed"] else [gi["ts_close"]])
            pauses.append(gi["pause"])
            msg_off.append(len(bits))
            pos_off.append(len(pos))
    for i, v in enumerate(vals):
        if v[0] > 0:
            go = gout[ex[i][1]]
            if go["is_msg"]:
                t = int(rows[i, 0])
                for k in range(v[0]):
                    b = 0 if t < 0 else (t >> (bps - 1 - k % bps)) & 1
                    bits[go["out_bits"] + ex[i][0] - go["bits_start"] + k] = b
                    if write_pos:
                        pos[go["out_pos"] + ex[i][0] - go["bits_start"] + k] = ex[i][2] + k * spb
    return (np.array(bits, dtype=np.uint8), np.array(msg_off, dtype=np.int64), np.array(pauses, dtype=np.int64),
            np.array(pos, dtype=np.int64), np.array(pos_off, dtype=np.int64))
