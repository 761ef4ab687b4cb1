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
