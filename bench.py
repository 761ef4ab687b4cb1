#!/usr/bin/env python3
"""bench.py -- IQ->bits throughput of the MI355X-native path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full pass of the hot path over one synthetic capture that is already resident in HBM:
IQ (complex64) -> demodulated signal (Signal.qad, materialised) -> pulse table -> bits / pauses /
bit_sample_pos, everything left in device memory.  Workload at N=1: BASELINE.json configs[1]
("1 GiB synthetic complex64 2-FSK @ 100 samples/symbol, single MI355X"); for N>1 every rank holds a
1 GiB sample-contiguous shard of one N-GiB capture (weak scaling, configs[3] at N=8).

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline      the dominant kernel (k_demod_runs_bp: demodulation + run segmentation) against HBM peak;
                achieved = algorithmic bytes (12 B/sample) / mean kernel time measured with HIP events
                on the launch stream inside the timed region
  cpu_baseline  the reference's own Cython kernels (oracle/_ref, built from /root/reference) when they
                are present, else the C port (oracle/), timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
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


def cpu_baseline(iq_host, sps, tol):
    """Reference CPU path on a bounded sample: afp_demod + grab_pulse_lens (+ _ppseq_to_bits port)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    import urh_oracle as oracle
    n = len(iq_host)
    kind = "port"
    afp, grab = oracle.afp_demod, oracle.grab_pulse_lens
    cores = 1
    if build_ref.built():
        sf, _, _ = build_ref.import_ref()
        afp = lambda a, nz, m, o: np.asarray(sf.afp_demod(a, nz, m, o))
        grab = lambda q, c, t, m, s, b, sp: np.asarray(sf.grab_pulse_lens(q, c, t, m, s, b, sp))
        kind = "reference"
        cores = os.cpu_count() or 1          # afp_demod is an OpenMP prange over all cores; the rest is serial
    warm = iq_host[: min(n, 1 << 20)]
    grab(afp(warm, 0.0, "FSK", 2), 0.0, tol, "FSK", sps, 1, 1.0)
    t0 = time.perf_counter()
    qad = afp(iq_host, 0.0, "FSK", 2)
    t1 = time.perf_counter()
    pp = grab(qad, 0.0, tol, "FSK", sps, 1, 1.0)
    t2 = time.perf_counter()
    oracle.ppseq_to_bits_flat(pp, sps, 1, True, 8)
    t3 = time.perf_counter()
    return {
        "value": round(n / (t3 - t0) / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": kind,
        "sample": f"first {n} samples of the same capture; stages: afp_demod {t1 - t0:.2f}s, grab_pulse_lens "
                  f"{t2 - t1:.2f}s ({kind} Cython), _ppseq_to_bits {t3 - t2:.2f}s (C port of the reference's pure-Python tail)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--segments", type=int, default=128, help="2^20-sample segments per GPU (128 = 1 GiB complex64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bits-only", action="store_true", help="do not materialise qad (8 B/sample variant)")
    ap.add_argument("--pipeline", action="store_true",
                    help="software-pipeline consecutive steps (hot kernel of step i+1 on the main stream while the tail of step i "
                         "runs on a second stream).  Default for --gpus N > 1, where the tail holds the boundary exchanges (three small "
                         "all-gathers with their cross-stream hand-overs: 0.46 -> 0.37 ms per step measured with a 1-rank RCCL group); "
                         "off for one GPU, where it buys 7 % throughput but stretches the dominant kernel the roofline line reports")
    ap.add_argument("--no-pipeline", action="store_true", help="never pipeline (see --pipeline)")
    args = ap.parse_args()

    import torch
    from urh_amd.pipeline import DemodParams, DevicePipeline
    from urh_amd.synth import fsk_capture

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # URH_BENCH_FORCE_SHARDED=1 (with torch.distributed.run --nproc-per-node 1) drives the sharded code path -- RCCL process
    # group, all-gathers, urhgpu_shard_* phases -- on a single GPU: a smoke test of the N > 1 plumbing on a 1-GPU box.
    force_sharded = os.environ.get("URH_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    sps, tol = 100, 5
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, tol, sps, 0.1, 8, True)
    # rank r holds segments [r*segments, (r+1)*segments) of the world*segments-segment capture
    if (world > 1 or force_sharded) and not args.no_pipeline:
        args.pipeline = True
    iq, tx_bits = fsk_capture(args.segments, dev, seed=1234, sps=sps, first_segment=rank * args.segments)
    n = iq.shape[0]
    if world > 1 or force_sharded:
        from urh_amd.shard_engine import GpuShardEngine
        from urh_amd.sharding import ShardedPipeline, TorchDistComm
        pipe = ShardedPipeline(GpuShardEngine(local_rank, pipelined=args.pipeline), TorchDistComm())
    else:
        pipe = DevicePipeline(local_rank, pipelined=args.pipeline)
    pipe.reserve(n, p)
    want_qad = not args.bits_only

    def step():
        return pipe.iq_to_bits(iq, p, want_qad=want_qad)

    for _ in range(args.warmup):
        res = step()
    pipe.ctx.join()
    torch.cuda.synchronize()
    # latency of ONE step with nothing overlapped (reported next to the pipelined throughput)
    lat = []
    for _ in range(5):
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        res = step()
        pipe.ctx.join()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t_l)
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

    # sanity on the last step's result: recovered bits vs transmitted bits (rank-local view)
    counts = res.host_counts()
    res.check_capacity()

    # N = 1 is timed WITHOUT software pipelining so that the roofline timing of the dominant kernel is undisturbed; N > 1 runs
    # pipelined (the tail holds the boundary exchanges).  For a like-for-like scaling comparison the N = 1 line also carries
    # the pipelined step time, measured separately after the timed region.
    pipelined_ms = None
    if world == 1 and not force_sharded and not args.pipeline and not args.no_pipeline:
        pp = DevicePipeline(local_rank, pipelined=True)
        pp.reserve(n, p)
        for _ in range(max(args.warmup, 1)):
            pp.iq_to_bits(iq, p, want_qad=want_qad)
        pp.ctx.join()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for _ in range(args.steps):
            rp = pp.iq_to_bits(iq, p, want_qad=want_qad)
        pp.ctx.join()
        torch.cuda.synchronize()
        pipelined_ms = (time.perf_counter() - tp) / args.steps * 1e3
        assert rp.host_counts() == counts
        pp.ctx.set_pipelined(False)

    if rank == 0:
        total_samples = n * world
        ms_per_step = dt / args.steps * 1e3
        value = total_samples * args.steps / dt / 1e6
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        bytes_per_sample = ALGO_BYTES_PER_SAMPLE if want_qad else 8
        achieved = (n * bytes_per_sample) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_src = (pmc_traffic("k_demod_runs_bp<0, 4, 1, true") if want_qad and n == 128 * SEG
                                else (None, None))
        out = {
            "metric": "Msamples/s IQ->bits (1 GiB complex64 2-FSK per GPU, qad materialised)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 1 GiB synthetic complex64 2-FSK @ 100 samples/symbol per GPU"
                       if world == 1 else f"configs[3]-style: {world} GiB complex64 2-FSK sharded sample-contiguously over {world} GPUs",
                       "samples_per_gpu": n, "samples_per_symbol": sps, "tolerance": tol, "noise_sigma": 0.05,
                       "outputs": "qad+ppseq+bits+pauses+bit_sample_pos" if want_qad else "ppseq+bits+pauses+bit_sample_pos",
                       "rows": counts[0], "messages": counts[1], "bits": counts[2],
                       "steps_pipelined": args.pipeline, "single_step_latency_ms": round(latency_ms, 4),
                       "pipelined_ms_per_step": round(pipelined_ms, 4) if pipelined_ms is not None else None},
            "roofline": {"bound": "hbm", "kernel": "k_demod_runs_bp", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "algorithmic_bytes": n * bytes_per_sample,
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_sample": bytes_per_sample,
                         "end_to_end_frac": round(n * bytes_per_sample / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            sample = n                      # the whole capture: the reference path takes about a second on the host cores
            out["cpu_baseline"] = cpu_baseline(iq[:sample].cpu().numpy(), sps, tol)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
