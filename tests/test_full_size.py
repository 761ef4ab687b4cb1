"""BASELINE.json configs[2] and configs[4] at FULL size (2^27 samples: the bytes SURVEY.md §8(d) configs 3 and 5 specify) against the
REAL reference compiled here (oracle/_ref: Cython fir_filter / afp_demod / costa_demod / grab_pulse_lens, Python detect_noise_level /
estimate / detect_center), every stage element for element -- the same records bench.py's `extra` carries, as assertions.
(configs[1], the 1 GiB 2-FSK capture: tests/test_gpu_parity.py::test_full_size_fsk_1gib_bit_exact.)"""
import argparse
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench_mod():
    import build_ref
    import ref_python
    if not (build_ref.built() and ref_python.available()):
        pytest.skip("oracle/_ref (the compiled reference + its Python sources) is not present on this box")
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _args():
    return argparse.Namespace(segments=128, no_cpu_baseline=False)


def test_full_size_config3_ook_fir_auto_noise_estimate_bits(bench_mod):
    """1 GiB OOK capture -> 64-tap complex FIR (Signal.filter_range semantics) with the magnitude chunk statistics fused ->
    detect_noise_level -> AutoInterpretation.estimate(noise, "OOK") -> bits sliced from the demodulated signal estimate left."""
    import torch
    from urh_amd.pipeline import DevicePipeline
    rec = bench_mod.extra_config3(DevicePipeline(0), torch.device("cuda", 0), _args())
    par = rec["parity"]
    assert par["fir_mismatches"] == 0, par
    assert par["noise_equal"] and par["estimate_equal"], par
    assert par["qad_mismatches"] == 0 and par["rows_equal"] and par["bits_pauses_positions_equal"], par
    assert par["messages_equal_transmitted_chips"] == 124 and rec["messages"] == 124, (par, rec["messages"])
    assert rec["unfused_ms"]["fused_result_equal"] and rec["for_comparison_ms"]["same_outputs"]
    assert par["bit_exact"]
    torch.cuda.empty_cache()


def test_full_size_config5_psk_costas_center_bits(bench_mod):
    """1 GiB 4-PSK: Costas loop (order 4) from sample 1 on (the reference leaves sample 0 uninitialised), detect_center, and the
    pulse table / bits / pauses / positions for the detected center and for center 0."""
    import torch
    from urh_amd.pipeline import DevicePipeline
    rec = bench_mod.extra_config5(DevicePipeline(0), torch.device("cuda", 0), _args())
    par = rec["parity"]
    assert par["qad_mismatches_from_index_1"] == 0, par
    assert par["center_equal"], par
    for tag in ("i_auto_center", "ii_center_0"):
        assert par[tag + "_rows_equal"] and par[tag + "_bits_pauses_positions_equal"], (tag, par)
    assert par["bit_exact"]
    assert rec["costas_chunks"]["speculative_hit_rate"] > 0.99, rec["costas_chunks"]
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant,upload", [("2", False), ("2b", False), ("2", True)])
def test_full_size_streamed_single_capture_equals_reference(oracle, variant, upload):
    """ONE 1 GiB capture through urhgpu_stream_* on an idle GPU -- the tail in segments beside the hot kernel, every segment's share of
    the compact blob stored straight into pinned host memory; upload: the capture starts on the host and is demodulated piece by piece
    as it lands -- equals the real reference (oracle/_ref) element for element: demodulated signal (uint32 view), pulse table, bits,
    pauses, message offsets, bit_sample_pos (variant 2b: the bursty capture, 128 messages)."""
    import ctypes as C

    import numpy as np
    import torch
    from test_gpu_parity import full_size_reference
    from urh_amd import _lib
    from urh_amd.pipeline import DemodParams, DevicePipeline
    from urh_amd.synth import spec_fsk_capture
    if variant == "2":
        iq, _ = spec_fsk_capture(128, "cuda:0")
        p = DemodParams("FSK", 1, 0.0, 0.0, 1.0, 5, 100, 0.1, 8, True)
    else:
        iq, _ = spec_fsk_capture(128, "cuda:0", seg_len=1 << 20, sps=100, n_symbols=10465)
        p = DemodParams("FSK", 1, 0.2, 0.0, 1.0, 5, 100, 0.1, 8, True)
    n = iq.shape[0]
    assert n == 1 << 27
    pipe = DevicePipeline(0, pipelined=True)
    st = pipe.stream(n, p, want_qad=True, want_pos=True)
    host = iq.cpu().numpy()
    if upload:
        pinned = torch.from_numpy(host).pin_memory()
        iq.zero_()
        torch.cuda.synchronize()
        assert st.push_upload(pinned, iq) is None
    else:
        assert st.push(iq) is None
    (r,) = st.flush()
    stats = st.stats()
    assert stats["predicted_bytes"] == -1, stats               # the pass took the segmented route
    r.check()
    rows, bits, msg_off, pauses, pos, pos_off = r.ppseq(), r.bits(), r.msg_off.copy(), r.pauses.copy(), r.bit_sample_pos(), r.pos_offsets()
    got_qad = np.empty(n, np.float32)
    _lib.check(_lib.load().urhgpu_memcpy_to_host(pipe.ctx.handle, C.c_void_p(r.d_qad_ptr), got_qad.ctypes.data_as(C.c_void_p), n * 4))
    if upload:
        assert np.array_equal(iq.cpu().numpy().view(np.uint32), host.view(np.uint32)), "the device buffer does not hold the uploaded capture"
    st.close()
    del iq
    torch.cuda.empty_cache()
    qad, pp, flat = full_size_reference(host, p, oracle)
    assert int((got_qad.view(np.uint32) != qad.view(np.uint32)).sum()) == 0
    assert np.array_equal(rows, pp), (len(rows), len(pp))
    for name, a, b in zip(("bits", "msg_off", "pauses", "pos", "pos_off"), (bits, msg_off, pauses, pos, pos_off), flat):
        assert np.array_equal(a, b), (name, len(a), len(b))
    assert len(pauses) == (1 if variant == "2" else 128)
