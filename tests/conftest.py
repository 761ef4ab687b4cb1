import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    for k in ("modulation_type", "kat", "kat_mode", "note"):
        g[k] = str(g[k])
    for k in ("bits_per_symbol", "tolerance", "samples_per_symbol", "pause_threshold"):
        g[k] = int(g[k])
    for k in ("noise_threshold", "center", "center_spacing", "costas_loop_bandwidth"):
        g[k] = float(g[k])
    return g


@pytest.fixture(scope="session")
def oracle():
    import urh_oracle
    urh_oracle.lib()
    return urh_oracle


def synth_fsk(n, sps=100, seed=0, noise=0.05, pause_every=0, pause_len=0, dtype=np.float32, deviation_hz=20e3):
    """Small seeded 2-FSK capture (continuous phase, +-deviation_hz @ 1 MS/s) with AWGN and optional silent gaps."""
    rng = np.random.default_rng(seed)
    nsym = n // sps + 1
    bits = rng.integers(0, 2, nsym)
    f = np.repeat(np.where(bits == 1, deviation_hz, -deviation_hz), sps)[:n]
    phase = np.cumsum(2 * np.pi * f / 1e6)
    iq = np.stack([np.cos(phase), np.sin(phase)], axis=1)
    if pause_every:
        for a in range(pause_every, n, pause_every + pause_len):
            iq[a:a + pause_len] = 0
    iq = iq + noise * rng.standard_normal((n, 2))
    if np.dtype(dtype) == np.float32:
        return iq.astype(np.float32)
    info = np.iinfo(dtype)
    scale = (info.max - info.min) / 2 * 0.7
    off = (info.max + info.min + 1) / 2
    return np.clip(np.round(iq * scale + off), info.min, info.max).astype(dtype)
