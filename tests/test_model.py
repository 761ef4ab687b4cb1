"""The chunk/tile algorithm of the HIP kernels (tests/model_runs.py) equals the oracle's serial state
machine on randomised captures with tiny tiles, so every boundary case (runs straddling tiles and
chunks, tentative first records, pending short runs, ASK merging, message groups) is exercised."""
import numpy as np

import model_runs as m


def _signal(rng, n, mod, bps, noise_val):
    levels = rng.choice([-1.0, -0.3, 0.3, 1.0] if bps == 2 else [-0.5, 0.5], size=n // max(1, int(rng.integers(1, 30))) + 1)
    x = np.repeat(levels, n // len(levels) + 1)[:n].astype(np.float32)
    x[rng.random(n) < rng.choice([0, 0.02, 0.1, 0.3])] *= -1
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n))
        x[a:min(n, a + int(rng.integers(1, 80)))] = noise_val
    if mod == "ASK":
        x = np.abs(x)
    return x


def test_model_equals_oracle(oracle):
    rng = np.random.default_rng(1)
    for it in range(600):
        n = int(rng.integers(1, 700))
        mod = ["ASK", "FSK", "PSK"][it % 3]
        bps = int(rng.integers(1, 3))
        tol = int(rng.choice([0, 1, 2, 3, 5, 7, 20, 100]))
        sps = int(rng.choice([3, 8, 20, 100]))
        noise_val = oracle.noise_for_mod_type(mod)
        x = _signal(rng, n, mod, bps, noise_val)
        center, spacing = (0.0 if mod != "ASK" else 0.4), 0.6
        thr = oracle.get_center_thresholds(center, spacing, 2 ** bps)
        ref = oracle.grab_pulse_lens(x, center, tol, mod, sps, bps, spacing)
        got = m.grab_pulse_lens_model(x, thr, np.float32(noise_val), tol, mod == "ASK", sps,
                                      tile=int(rng.choice([16, 32, 64])), span=int(rng.choice([4, 8])),
                                      chunk_tiles=int(rng.choice([1, 2, 3])))
        assert np.array_equal(ref, got), (it, n, mod, bps, tol, sps)
        for pt in (0, 1, 8):
            fb = oracle.ppseq_to_bits_flat(ref, sps, bps, True, pt)
            mb = m.ppseq_to_bits_model(ref, sps, bps, True, pt)
            assert all(np.array_equal(a, b) for a, b in zip(fb, mb)), (it, pt)
