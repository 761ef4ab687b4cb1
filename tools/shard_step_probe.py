"""tools/shard_step_probe.py [fir|plain|host] [steps]: the sharded engine's step on ONE rank without a process group (an identity all-gather).
  fir    the FIR-halo variant of configs[3]: 64-tap FIR -> sharded IQ->bits (positions on), device-resident, K steps; the filter alone; the pass alone
  plain  the sharded IQ->bits step with its blob on the host (the N > 1 headline loop)
  host   `plain`, with the host's time inside every call of a step added up (is the loop bound by the host?)
Run under rocprofv3 --kernel-trace for the timeline (tools/timeline.py --all)."""
import os
import sys
import time
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from urh_amd.pipeline import DemodParams
from urh_amd.shard_engine import GpuShardEngine
from urh_amd.sharding import ShardedPipeline
from urh_amd.synth import spec_fir_taps, spec_fsk_capture


class OneRank:
    rank, world = 0, 1

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_gather_start(self, t):
        return lambda: t.unsqueeze(0)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "fir"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    iq, _ = spec_fsk_capture(128, dev)
    n = iq.shape[0]
    p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, False)
    tuning = {k[9:].lower(): int(v) for k, v in os.environ.items() if k.startswith("URH_TUNE_")}       # URH_TUNE_<KEY>=<value>: urhgpu_ctx_set_tuning
    pipe = ShardedPipeline(GpuShardEngine(0, pipelined=True, tuning=tuning), OneRank())
    pipe.reserve(n, p)
    e = pipe.engine

    def timed(fn, k):
        for _ in range(max(10, k)):
            fn()
        pipe.ctx.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        pipe.ctx.join(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3

    if what == "fir":
        taps = torch.from_numpy(spec_fir_taps().view("float32").reshape(-1, 2).copy()).to(dev)
        p_pos = replace(p, write_bit_sample_pos=True)
        e.host_results = False

        def step():
            x, fh = pipe.fir_filter(iq, taps, left_raw=None, want_halo=True)
            return pipe.iq_to_bits(x, p_pos, want_qad=True, halo_given=True, left_halo=fh)
        x0 = pipe.fir_filter(iq, taps)
        print(f"fir-halo step {timed(step, steps):.4f} ms | filter alone {timed(lambda: pipe.fir_filter(iq, taps), steps):.4f} | "
              f"pass alone (positions) {timed(lambda: pipe.iq_to_bits(x0, p_pos, want_qad=True, halo_given=True), steps):.4f} | "
              f"pass alone (no positions) {timed(lambda: pipe.iq_to_bits(x0, p, want_qad=True, halo_given=True), steps):.4f}")
        return
    e.host_results = True
    spent = {}
    if what == "host":
        def wrap(obj, name):
            f = getattr(obj, name)

            def g(*a, **k):
                t0 = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    spent[name] = spent.get(name, 0.0) + time.perf_counter() - t0
            setattr(obj, name, g)
        for nm in ("runs_launch", "runs", "rows", "bits_prepare", "bits_finish", "_setup", "_queue_host_copy", "_finish_host_copy"):
            wrap(e, nm)

    def host_steps(k):
        pend, last = [], None
        for _ in range(k):
            pend.append(pipe.iq_to_bits(iq, p, want_qad=True, halo_given=True))
            if len(pend) > 2:
                last = pend.pop(0).host()
        for r in pend:
            last = r.host()
        return last
    for _ in range(5):
        host_steps(20)
    torch.cuda.synchronize()
    spent.clear()
    t0 = time.perf_counter()
    host_steps(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"sharded step with its blob on the host: {dt:.4f} ms")
    if spent:
        print("host time inside the calls, us per step (runs_launch holds the bounded run-ahead's wait; _setup is inside runs_launch):")
        for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
            print(f"    {k:20s} {v / steps * 1e6:8.1f}")


if __name__ == "__main__":
    main()
