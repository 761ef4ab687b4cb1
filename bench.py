#!/usr/bin/env python3
"""bench.py -- IQ->bits throughput of the MI355X-native path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full pass of the hot path over one synthetic capture that is already resident in HBM:
IQ (complex64) -> demodulated signal (Signal.qad, materialised) -> pulse table -> bits / pauses /
bit_sample_pos, everything left in device memory.  Workload at N=1: BASELINE.json configs[1]
("1 GiB synthetic complex64 2-FSK @ 100 samples/symbol, single MI355X") on the bytes SURVEY.md §8(d) config 2
specifies (128 segments of 2^20 samples: numpy-seeded bits through modulate_c + numpy-seeded AWGN, see
urh_amd/synth.py:spec_fsk_capture); for N>1 every rank holds a 1 GiB sample-contiguous shard of one N-GiB capture
(segments 128*rank ..., weak scaling, configs[3] at N=8).

`python bench.py --gpus N` with N > 1 starts its own ranks (it re-executes itself under torch.distributed.run,
one process per GPU over RCCL) unless it already runs inside such a launch (WORLD_SIZE set).

Prints ONE JSON line on rank 0 (see the driver contract) with extra objects:
  roofline      the dominant kernel (k_demod_runs_bp: demodulation + run segmentation) against HBM peak;
                achieved = algorithmic bytes (12 B/sample) / mean kernel time measured with HIP events
                attached to the kernel's dispatch on its launch stream, inside the timed region
  parity        the LAST timed step's outputs (qad as uint32, pulse table, bits, pauses, offsets, bit_sample_pos) of the full
                2^27-sample capture compared element for element with the CPU reference on the same bytes
  cpu_baseline  the reference's own path on this box's host cores (oracle/_ref = the reference's Cython modules compiled
                from /root/reference, plus its pure-Python tail when the staged sources are present), else the C port
  extra         other BASELINE configurations at full size (--extra): configs[2] OOK+FIR+auto noise, configs[4] 4-PSK Costas
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEG = 1 << 20
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALGO_BYTES_PER_SAMPLE = 12       # 8 B complex64 read + 4 B float32 qad write


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary under profiles/
    (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, corrected as tools/prof_collect.py documents); the counters
    cannot be collected inside this process, so this is the figure of the profiled run of the SAME command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fsk_1gib_pmc.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for name, rec in d.items():
            if kernel_substr in name and "hbm_traffic_bytes_per_launch" in rec:
                return int(rec["hbm_traffic_bytes_per_launch"]["total"]), os.path.basename(f)
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _set_omp_threads(n):
    """omp_set_num_threads on the OpenMP runtime the reference's prange uses (libgomp, already loaded with the module)."""
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def cpu_reference(iq_host, p, want_outputs=True):
    """The CPU side on the whole capture: (qad, ppseq, flat bits) for the parity check and the cpu_baseline record.

    With oracle/_ref (the reference's own Cython modules, compiled from /root/reference): afp_demod (OpenMP prange over all host
    cores) + grab_pulse_lens, timed per stage with OMP_NUM_THREADS unset and = 1; with the staged Python sources also the
    reference's end-to-end ProtocolAnalyzer.get_protocol_from_signal() (incl. its pure-Python _ppseq_to_bits), median of 3 after
    a warm-up on one segment.  Without oracle/_ref: the C restatement (kind "port", 1 thread)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    import ref_python
    import urh_oracle as oracle
    n = len(iq_host)
    order = 2 ** p.bits_per_symbol
    mod, sps, tol = p.modulation_type, p.samples_per_symbol, p.tolerance
    rec = {"unit": "Msamples/s", "cpu_model": cpu_model(), "host_cores": os.cpu_count() or 1}
    if not build_ref.built():
        t0 = time.perf_counter()
        qad = oracle.afp_demod(iq_host, p.noise_threshold, mod, order)
        t1 = time.perf_counter()
        pp = oracle.grab_pulse_lens(qad, p.center, tol, mod, sps, p.bits_per_symbol, p.center_spacing)
        t2 = time.perf_counter()
        flat = oracle.ppseq_to_bits_flat(pp, sps, p.bits_per_symbol, True, p.pause_threshold)
        t3 = time.perf_counter()
        rec.update({"value": round(n / (t3 - t0) / 1e6, 2), "cores": 1, "kind": "port",
                    "sample": f"all {n} samples of the same capture through the C restatement (oracle/urh_oracle.c): afp_demod {t1 - t0:.2f}s, "
                              f"grab_pulse_lens {t2 - t1:.2f}s, _ppseq_to_bits {t3 - t2:.2f}s"})
        return rec, (qad, pp, flat)
    sf, _, _ = build_ref.import_ref()
    cores = os.cpu_count() or 1
    afp = lambda a: np.asarray(sf.afp_demod(a, p.noise_threshold, mod, order, p.costas_loop_bandwidth))          # noqa: E731
    grab = lambda q: np.asarray(sf.grab_pulse_lens(q, p.center, tol, mod, sps, p.bits_per_symbol, p.center_spacing))   # noqa: E731
    grab(afp(iq_host[:SEG]))                        # warm-up
    t0 = time.perf_counter()
    qad = afp(iq_host)
    t1 = time.perf_counter()
    pp = grab(qad)
    t2 = time.perf_counter()
    flat = oracle.ppseq_to_bits_flat(pp, sps, p.bits_per_symbol, True, p.pause_threshold)       # for the parity check
    t3 = time.perf_counter()
    stages = {"afp_demod_s": round(t1 - t0, 3), "grab_pulse_lens_s": round(t2 - t1, 3), "ppseq_to_bits_c_port_s": round(t3 - t2, 3)}
    # the same two Cython stages on ONE thread (afp_demod is the only threaded stage)
    if _set_omp_threads(1):
        ta = time.perf_counter()
        q1 = afp(iq_host)
        tb = time.perf_counter()
        stages["afp_demod_1thread_s"] = round(tb - ta, 3)
        del q1
        _set_omp_threads(cores)
    value_s = t3 - t0
    kind_note = "Cython stages of the real reference + C port of its pure-Python tail"
    if ref_python.available():
        try:
            ref_python.setup()
            from urh.signalprocessing.IQArray import IQArray
            from urh.signalprocessing.ProtocolAnalyzer import ProtocolAnalyzer
            from urh.signalprocessing.Signal import Signal

            def e2e(arr):
                s = Signal("")
                s.iq_array = IQArray(arr)
                s.modulation_type = mod
                s.bits_per_symbol = p.bits_per_symbol
                s.noise_threshold = p.noise_threshold
                s.center, s.center_spacing, s.tolerance = p.center, p.center_spacing, tol
                s.samples_per_symbol, s.pause_threshold = sps, p.pause_threshold
                pa = ProtocolAnalyzer(s)
                ts = time.perf_counter()
                pa.get_protocol_from_signal()              # qad (afp_demod) -> grab_pulse_lens -> _ppseq_to_bits (pure Python) -> Messages
                return time.perf_counter() - ts, pa

            e2e(iq_host[:SEG])
            runs = sorted(e2e(iq_host)[0] for _ in range(3))
            stages["get_protocol_from_signal_s_runs"] = [round(x, 3) for x in runs]
            value_s = runs[1]
            _set_omp_threads(1)
            t1t, pa = e2e(iq_host)
            _set_omp_threads(cores)
            stages["get_protocol_from_signal_1thread_s"] = round(t1t, 3)
            rec["value_1thread"] = round(n / t1t / 1e6, 2)
            if want_outputs:           # the reference's own Python tail agrees with the C port used for the parity check
                ref_bits = "".join(m.plain_bits_str for m in pa.messages)
                stages["python_tail_equals_c_port"] = bool(ref_bits == "".join(map(str, flat[0].tolist()))) if len(flat[0]) < 50_000_000 else None
            kind_note = "the reference's own ProtocolAnalyzer.get_protocol_from_signal() end to end (Cython afp_demod + grab_pulse_lens, pure-Python _ppseq_to_bits), median of 3"
        except Exception as e:           # noqa: BLE001  (the baseline must not take the benchmark down)
            stages["python_path_error"] = repr(e)[:200]
    rec.update({"value": round(n / value_s / 1e6, 2), "cores": cores, "kind": "reference",
                "sample": f"all {n} samples of the same capture; {kind_note}; OMP_NUM_THREADS unset ({cores} threads) and = 1 (value_1thread)",
                "stages": stages})
    return rec, (qad, pp, flat)


def parity_record(res, ref_out, tx_bits, kind):
    """Element-for-element comparison of the last timed step's device outputs with the CPU reference's on the same bytes."""
    import numpy as np
    qad, pp, flat = ref_out
    got_qad = res.qad.cpu().numpy() if res.qad is not None else None
    rows = res.ppseq()
    got = res.flat()
    names = ("bits", "msg_off", "pauses", "bit_sample_pos", "pos_off")
    rec = {"against": "oracle/_ref (the reference's Cython afp_demod + grab_pulse_lens; tail: C port pinned on the reference's Python)"
           if kind == "reference" else "oracle/ C restatement", "samples": int(len(qad))}
    if got_qad is not None:
        rec["qad_mismatches"] = int((got_qad.view(np.uint32) != qad.view(np.uint32)).sum())
    rec["rows"] = int(len(pp))
    rec["rows_equal"] = bool(np.array_equal(rows, pp))
    for k, name in enumerate(names):
        rec[name + "_equal"] = bool(np.array_equal(got[k], flat[k]))
    rec["n_bits"] = int(len(flat[0]))
    rec["n_messages"] = int(len(flat[2]))
    # sanity against the transmitter (not the parity criterion): bit k of a message starts at sample pos[k]
    bits, off, pauses, pos, poff = got
    errors = total = 0
    for m in range(len(pauses)):
        bp = pos[poff[m]:poff[m] + (off[m + 1] - off[m])]
        seg_of, sym_of = bp // SEG, (bp % SEG + 50) // 100
        ok = (sym_of < tx_bits.shape[1]) & (seg_of < tx_bits.shape[0])
        errors += int((bits[off[m]:off[m + 1]][ok] != tx_bits[seg_of[ok], sym_of[ok]]).sum())
        total += int(ok.sum())
    rec["bit_errors_vs_transmitted"] = errors
    rec["bits_compared_vs_transmitted"] = total
    rec["bit_exact"] = bool(rec.get("qad_mismatches", 0) == 0 and rec["rows_equal"] and all(rec[nm + "_equal"] for nm in names))
    return rec


def run_extras(pipe, dev, args):
    return []


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) from a bare interpreter: become `python -m torch.distributed.run ... bench.py ...`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--segments", type=int, default=128, help="2^20-sample segments per GPU (128 = 1 GiB complex64)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU reference (then no parity record either)")
    ap.add_argument("--bits-only", action="store_true", help="do not materialise qad (8 B/sample variant)")
    ap.add_argument("--torch-capture", action="store_true", help="round-1 capture (torch RNG, clean fp64 phase ramp) instead of the §8(d) bytes")
    ap.add_argument("--extra", action="store_true", help="also run configs[2] and configs[4] at full size (adds about a minute)")
    ap.add_argument("--fir-halo", action="store_true", help="N > 1: prepend the 64-tap FIR with halo exchange (configs[3] 'FIR-halo' variant)")
    ap.add_argument("--pipeline", action="store_true",
                    help="software-pipeline consecutive steps (hot kernel of step i+1 on the main stream while the tail of step i "
                         "runs on a second stream).  Default for --gpus N > 1, where the tail holds the boundary exchanges; "
                         "off for one GPU, where it stretches the dominant kernel the roofline line reports")
    ap.add_argument("--no-pipeline", action="store_true", help="never pipeline (see --pipeline)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    from urh_amd.synth import fsk_capture, spec_fsk_capture

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # URH_BENCH_FORCE_SHARDED=1 (with torch.distributed.run --nproc-per-node 1) drives the sharded code path -- RCCL process
    # group, all-gathers, urhgpu_shard_* phases -- on a single GPU: a smoke test of the N > 1 plumbing on a 1-GPU box.
    force_sharded = os.environ.get("URH_BENCH_FORCE_SHARDED") == "1"
    sharded = world > 1 or force_sharded
    if sharded:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    sps, tol = 100, 5
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, tol, sps, 0.1, 8, True)
    # rank r holds segments [r*segments, (r+1)*segments) of the world*segments-segment capture
    if sharded and not args.no_pipeline:
        args.pipeline = True
    if args.torch_capture:
        iq, tx_bits = fsk_capture(args.segments, dev, seed=1234, sps=sps, first_segment=rank * args.segments)
        tx_bits = tx_bits.cpu().numpy()
        capture = "torch RNG (round-1 generator)"
    else:
        iq, tx_bits = spec_fsk_capture(args.segments, dev, first_segment=rank * args.segments, sps=sps)
        capture = "SURVEY 8(d) config 2 bytes: modulate_c segments (default_rng(1234+k) bits) + default_rng(5678+k) AWGN 0.05"
    n = iq.shape[0]
    fir_taps = None
    if sharded:
        from urh_amd.shard_engine import GpuShardEngine
        from urh_amd.sharding import ShardedPipeline, TorchDistComm
        pipe = ShardedPipeline(GpuShardEngine(local_rank, pipelined=args.pipeline), TorchDistComm())
        if args.fir_halo:
            from urh_amd.synth import spec_fir_taps
            fir_taps = torch.from_numpy(spec_fir_taps().view("float32").copy()).to(dev)
    else:
        pipe = DevicePipeline(local_rank, pipelined=args.pipeline)
    pipe.reserve(n, p)
    want_qad = not args.bits_only

    def step():
        x = pipe.fir_filter(iq, fir_taps) if fir_taps is not None else iq
        return pipe.iq_to_bits(x, p, want_qad=want_qad)

    for _ in range(args.warmup):
        res = step()
    pipe.ctx.join()
    torch.cuda.synchronize()
    # latency of ONE step with nothing overlapped, and the same plus the D2H copy of the compact outputs (pulse table, bits,
    # pauses, offsets, bit_sample_pos -- SURVEY 8(d)'s timing window; qad stays in HBM)
    lat, lat_d2h = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        res = step()
        pipe.ctx.join()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t_l)
    d2h_bytes = None
    if not sharded:
        for _ in range(5):
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            res = step()
            rows_h = res.ppseq()
            flat_h = res.flat()
            lat_d2h.append(time.perf_counter() - t_l)
        d2h_bytes = int(rows_h.nbytes + sum(x.nbytes for x in flat_h))
    latency_ms = min(lat) * 1e3
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    pipe.ctx.profile_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    pipe.ctx.join()                      # the stream waits for the last step's tail
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = pipe.ctx.profile_end()
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    counts = res.host_counts()
    res.check_capacity()

    # N = 1 is timed WITHOUT software pipelining so that the roofline timing of the dominant kernel is undisturbed; N > 1 runs
    # pipelined (the tail holds the boundary exchanges).  For a like-for-like scaling comparison the N = 1 line also carries
    # the pipelined step time, measured separately after the timed region.
    pipelined_ms = None
    if not sharded and not args.pipeline and not args.no_pipeline:
        pp = DevicePipeline(local_rank, pipelined=True)
        pp.reserve(n, p)
        for _ in range(max(args.warmup, 1)):
            pp.iq_to_bits(iq, p, want_qad=want_qad, slot=1)
        pp.ctx.join()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(args.steps):
            rp = pp.iq_to_bits(iq, p, want_qad=want_qad, slot=1)
        pp.ctx.join()
        torch.cuda.synchronize()
        pipelined_ms = (time.perf_counter() - tp) / args.steps * 1e3
        assert rp.host_counts() == counts
        pp.ctx.set_pipelined(False)
        del pp, rp

    ranks_info = None
    if dist:
        info = [None] * world
        dist.all_gather_object(info, {"rank": rank, "device": torch.cuda.get_device_name(dev), "index": local_rank,
                                      "rows": counts[0], "bits": counts[2]})
        ranks_info = info

    if rank == 0:
        total_samples = n * world
        ms_per_step = dt / args.steps * 1e3
        value = total_samples * args.steps / dt / 1e6
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        bytes_per_sample = ALGO_BYTES_PER_SAMPLE if want_qad else 8
        achieved = (n * bytes_per_sample) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = (pmc_traffic("k_demod_runs_bp<0, 4, 1, true") if want_qad and n == 128 * SEG
                                else (None, None))
        e2e_frac = n * bytes_per_sample / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "Msamples/s IQ->bits (1 GiB complex64 2-FSK per GPU, qad materialised)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 1 GiB synthetic complex64 2-FSK @ 100 samples/symbol per GPU"
                       if world == 1 else f"configs[3]-style: {world} GiB complex64 2-FSK sharded sample-contiguously over {world} GPUs"
                       + (" with the 64-tap FIR (63-sample halo exchange) in front" if fir_taps is not None else ""),
                       "capture": capture,
                       "samples_per_gpu": n, "samples_per_symbol": sps, "tolerance": tol, "noise_sigma": 0.05,
                       "outputs": "qad+ppseq+bits+pauses+bit_sample_pos" if want_qad else "ppseq+bits+pauses+bit_sample_pos",
                       "rows": counts[0], "messages": counts[1], "bits": counts[2],
                       "steps_pipelined": args.pipeline, "single_step_latency_ms": round(latency_ms, 4),
                       "single_step_plus_d2h_ms": round(min(lat_d2h) * 1e3, 4) if lat_d2h else None, "d2h_bytes": d2h_bytes,
                       "pipelined_ms_per_step": round(pipelined_ms, 4) if pipelined_ms is not None else None,
                       "rccl_world_size": world if dist else None, "ranks": ranks_info},
            "roofline": {"bound": "hbm", "kernel": "k_demod_runs_bp", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "algorithmic_bytes": n * bytes_per_sample,
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_sample": bytes_per_sample,
                         "end_to_end_frac": round(e2e_frac, 4),
                         "end_to_end_plus_d2h_frac": round(n * bytes_per_sample / min(lat_d2h) / 1e9 / HBM_PEAK_GBS, 4) if lat_d2h else None},
        }
        if not args.no_cpu_baseline and world == 1 and not force_sharded:
            host = iq.cpu().numpy()
            rec, ref_out = cpu_reference(host, p)
            out["cpu_baseline"] = rec
            out["parity"] = parity_record(res, ref_out, tx_bits, rec["kind"])
            del host, ref_out
        if args.extra and world == 1 and not force_sharded:
            del iq
            out["extra"] = run_extras(pipe, dev, args)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
