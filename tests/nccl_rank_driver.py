#!/usr/bin/env python3
"""One rank of a REAL `nccl` (= RCCL) process group driving the HIP shard engine (launched by tests/test_reference_dropin.py
::test_gpu_shard_engine_over_rccl_group through torch.distributed.run; world size = however many ranks were started, 1 on the
1-GPU test box).  Every rank holds a sample-contiguous shard of one capture, runs the 64-tap FIR with halo exchange and the
sharded IQ->bits pass (all-gathers over RCCL), gathers the pieces on rank 0, which compares the stitched result with the oracle
on the whole capture -- bit-exact -- and prints "RCCL_SHARD_OK <world>"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    from conftest import synth_fsk
    from urh_amd.pipeline import DemodParams
    from urh_amd.shard_engine import GpuShardEngine
    from urh_amd.sharding import RcclComm, ShardedPipeline, TorchDistComm, shard_bounds, stitch
    from urh_amd.synth import spec_fir_taps
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    n = (3 << 20) + 4097 * world
    iq = synth_fsk(n, sps=100, seed=77, noise=0.05, pause_every=700_000, pause_len=40_000)
    taps = spec_fir_taps()
    a, b = shard_bounds(n, world)[rank]
    shard = torch.from_numpy(iq[a:b]).to(dev)
    rccl = RcclComm(None)                                     # the direct communicator must come up here (no fallback in the test)
    for pipelined, comm in ((False, TorchDistComm()), (True, TorchDistComm()), (False, rccl), (True, rccl)):
        pipe = ShardedPipeline(GpuShardEngine(local, pipelined=pipelined, host_results=pipelined), comm)   # pipelined: + the blob on the host
        assert pipe.world == world and pipe.rank == rank
        p = DemodParams("FSK", 1, 0.1, 0.0, 1.0, 5, 100, 0.1, 8, True)
        d_taps = torch.from_numpy(taps.view(np.float32).reshape(-1, 2).copy()).to(dev)
        for it in range(4 if pipelined else 1):
            filt = pipe.fir_filter(shard, d_taps)
            # odd passes: the halo comes with the shard (here: gathered by hand beforehand), two exchanges per pass
            given = it % 2 == 1
            left = None
            if given:
                tails = comm.all_gather(filt[-2:].contiguous())
                left = tails[rank - 1].clone() if rank > 0 else None
            res = pipe.iq_to_bits(filt, p, want_qad=True, pos_base=a, n_total=n, halo_given=given, left_halo=left)
        pipe.ctx.join()
        torch.cuda.synchronize()
        piece = res.piece()
        if pipelined:
            h = res.host().check()
            assert np.array_equal(h.ppseq(), piece["rows"]) and np.array_equal(h.bits(), piece["bits"]), "host blob differs from the device outputs"
            assert np.array_equal(h.pauses, piece["pauses"]) and np.array_equal(h.bit_sample_pos(), piece["pos"]), "host blob: pauses / positions"
        piece["qad"] = res.qad.cpu().numpy()
        piece["filt"] = filt.cpu().numpy()
        pieces = [None] * world
        dist.all_gather_object(pieces, piece)
        if rank == 0:
            import urh_oracle as oracle
            want_f = oracle.fir_filter(np.ascontiguousarray(iq).view(np.complex64).reshape(-1), taps).view(np.float32).reshape(-1, 2)
            got_f = np.concatenate([pc["filt"] for pc in pieces])
            assert np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32)), "sharded FIR differs"
            qad = oracle.afp_demod(want_f, 0.1, "FSK", 2)
            pp = oracle.grab_pulse_lens(qad, 0.0, 5, "FSK", 100, 1, 1.0)
            flat = oracle.ppseq_to_bits_flat(pp, 100, 1, True, 8)
            got = stitch(pieces)
            assert np.array_equal(np.concatenate([pc["qad"] for pc in pieces]).view(np.uint32), qad.view(np.uint32)), "qad differs"
            assert np.array_equal(got[0], pp), "pulse table differs"
            for k in range(5):
                assert np.array_equal(got[1 + k], flat[k]), k
        if pipelined:
            pipe.ctx.set_pipelined(False)
    dist.barrier()
    rccl.close()
    if rank == 0:
        print(f"RCCL_SHARD_OK {world} backend={dist.get_backend()}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
