"""Boundary proof: the reference's OWN demodulation tests pass with this library bound underneath the
reference's Signal / ProtocolAnalyzer through INTEGRATION.md §1's monkeypatch, and the host-threading contract holds
(any thread, own context per thread, spawn-safe with lazy HIP initialisation in the child)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synth_fsk

pytestmark = pytest.mark.gpu


def _run_driver(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_driver.py"), *args], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if out.get("unavailable"):
        pytest.skip("oracle/_ref (compiled reference + staged Python sources) is not present on this box")
    return out


def test_reference_test_demodulations_on_gpu_functions():
    """/root/reference/tests/test_demodulations.py (:14-27 ASK, :29-40 ASK tol 0, :42-53 FSK exact 177 bits, :55-72 modulate ->
    demodulate, :74-87 PSK, :89-120 4-PSK clean + noisy, :122-135 4-FSK), unmodified, with afp_demod / grab_pulse_lens /
    get_center_thresholds / modulate_c coming from liburhgpu.so."""
    out = _run_driver("test_demodulations")
    rec = out["per_module"]["tests.test_demodulations"]
    assert rec["ran"] == 7 and rec["failures"] == 0 and rec["errors"] == 0, out["details"]
    # the GPU functions did the work: every test demodulates and slices, three of them modulate
    c = out["calls"]
    assert c["signal_functions.afp_demod"] >= 8 and c["signal_functions.grab_pulse_lens"] >= 8 and c["signal_functions.modulate_c"] >= 3, c


def test_reference_hot_path_tests_on_gpu_functions():
    """The reference's other headless hot-path tests (SURVEY.md probe table), unmodified but for the import of
    get_path_for_data_file, with urh.cythonext.signal_functions / auto_interpretation / util rebound to liburhgpu.so:
    tests/auto_interpretation/*.py (AutoInterpretation.estimate, detect_center, detect_noise_level, segmentation, OOK merge, bit
    length, tolerance, modulation detection on the reference's captures), tests/test_protocol_analyzer.py:11-61,
    tests/test_iq_array.py, tests/test_modulator.py, and the FIR known-answer test of tests/test_filter.py:20-31."""
    out = _run_driver()
    assert out["failures"] == 0 and out["errors"] == 0, "\n".join(out["details"])[-6000:]
    assert out["ran"] == 75 and out["skipped"] == 0, out["per_module"]
    c = out["calls"]
    for key in ("signal_functions.afp_demod", "signal_functions.grab_pulse_lens", "signal_functions.fir_filter", "signal_functions.modulate_c",
                "auto_interpretation.segment_messages_from_magnitudes", "auto_interpretation.get_threshold_divisor_histogram",
                "auto_interpretation.merge_plateaus", "auto_interpretation.get_plateau_lengths", "auto_interpretation.median_filter",
                "util.minmax", "util.get_magnitudes"):
        assert c[key] > 0, (key, c)


def test_hook_binds_on_a_gpu_box():
    """urh_amd/urh_hook.install() -- what a maintainer puts into src/urh/cythonext/__init__.py -- binds the
    library where a GPU is usable (here); without one it keeps the Cython functions (tests/test_oracle.py::test_hook_keeps_cython_without_gpu)."""
    out = _run_driver("--hook", "test_demodulations")
    assert out["hook"]["installed"] is True, out["hook"]
    rec = out["per_module"]["tests.test_demodulations"]
    assert rec["ran"] == 7 and rec["failures"] == 0 and rec["errors"] == 0, out["details"]
    assert out["calls"]["signal_functions.afp_demod"] >= 8 and out["calls"]["signal_functions.grab_pulse_lens"] >= 8, out["calls"]


def _thread_job(k, out, errs):
    try:
        import urh_oracle as oracle
        from urh_amd import _lib, signal_functions as sf
        iq = synth_fsk(200_000 + 1000 * k, sps=50, seed=40 + k, noise=0.05)
        ctx = _lib.default_context()                 # thread-local
        for _ in range(5):
            qad = sf.afp_demod(iq, 0.0, "FSK", 2)
            pp = sf.grab_pulse_lens(qad, 0.0, 5, "FSK", 50)
        out[k] = (id(ctx), np.array_equal(qad.view(np.uint32), oracle.afp_demod(iq, 0.0, "FSK", 2).view(np.uint32)),
                  np.array_equal(pp, oracle.grab_pulse_lens(oracle.afp_demod(iq, 0.0, "FSK", 2), 0.0, 5, "FSK", 50, 1, 1.0)))
    except Exception as e:               # noqa: BLE001
        errs.append(repr(e))


def test_two_host_threads_own_contexts(oracle):
    """GUI thread + sniffer thread (ProtocolSniffer.py:68, 161-165): concurrent callers, one context each, both bit-exact."""
    import threading
    out, errs = {}, []
    ts = [threading.Thread(target=_thread_job, args=(k, out, errs)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert out[0][1] and out[0][2] and out[1][1] and out[1][2]
    assert out[0][0] != out[1][0], "each host thread must get its own context"


def _child(q, path):
    try:
        sys.path.insert(0, path)
        from urh_amd import signal_functions as sf     # HIP is initialised lazily, here in the child
        rng = np.random.default_rng(5)
        iq = rng.standard_normal((30_000, 2)).astype(np.float32)
        q.put(("ok", sf.afp_demod(iq, 0.0, "FSK", 2).tobytes()))
    except Exception as e:               # noqa: BLE001
        q.put(("err", repr(e)))


def test_spawned_child_process_lazy_init(oracle):
    """The band-pass worker is a multiprocessing.Process (SignalFrame.py:1553-1564; spawn start method in the reference's
    tests, QtTestCase.py:28-34): a spawned child initialises HIP on its own after the parent has already used the GPU."""
    import multiprocessing as mp
    from urh_amd import signal_functions as sf
    rng = np.random.default_rng(5)
    iq = rng.standard_normal((30_000, 2)).astype(np.float32)
    want = oracle.afp_demod(iq, 0.0, "FSK", 2)
    assert np.array_equal(sf.afp_demod(iq, 0.0, "FSK", 2).view(np.uint32), want.view(np.uint32))     # parent has a live context
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_child, args=(q, ROOT))
    pr.start()
    status, payload = q.get(timeout=300)
    pr.join(60)
    assert status == "ok", payload
    assert payload == want.tobytes()


def test_gpu_shard_engine_over_rccl_group():
    """The N > 1 code path as bench.py --gpus N runs it -- GpuShardEngine phases + TorchDistComm all-gathers over a real `nccl`
    (RCCL) process group, FIR halo included, plain and pipelined -- on as many ranks as this box has GPUs (1 on the test box),
    stitched result bit-exact against the oracle."""
    import socket
    import torch
    world = max(1, min(torch.cuda.device_count(), 8))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "nccl_rank_driver.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0 and f"RCCL_SHARD_OK {world} backend=nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
