"""Plot decimation (path_creator.create_path, path_creator.pyx:19-82): the oracle and the GPU pass against vertex arrays
produced by the REAL reference function (tests/golden/path/paths.npz, made by tests/golden/make_path_golden.py through
oracle/_ref and the PyQt6 stub's QDataStream), bit-exact including NaN / signed-zero behaviour and the unsigned wrap of
np.negative."""
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "path", "paths.npz")


def cases():
    g = np.load(GOLD)
    for name in g["names"]:
        name = str(name)
        start, end = (int(v) for v in g[name + "_args"])
        ranges = None if g[name + "_default_ranges"][0] else [tuple(r) for r in g[name + "_ranges"]]
        want = []
        k = 0
        while f"{name}_x{k}" in g:
            want.append((g[f"{name}_x{k}"], g[f"{name}_y{k}"]))
            k += 1
        yield name, g[name + "_samples"], start, end, ranges, want


def check(got, want, name):
    assert len(got) == len(want), name
    for (x, v), (wx, wy) in zip(got, want):
        assert np.array_equal(x.astype(np.float64), wx), name
        assert np.array_equal(np.negative(v).astype(np.float64), wy, equal_nan=True), name


def decode(b):
    n = int.from_bytes(b[:4], "big", signed=True)
    arr = np.frombuffer(b, dtype=[("c", ">i4"), ("x", ">f8"), ("y", ">f8")], count=n, offset=4)
    assert (arr["c"] == 1).all() and b[4 + 20 * n:] == bytes(8)
    return arr["x"].astype(np.float64), arr["y"].astype(np.float64)


def test_oracle_equals_reference_paths(oracle):
    from urh_amd.path_creator import path_bytes
    n = 0
    for name, samples, start, end, ranges, want in cases():
        got = oracle.create_path_arrays(samples, start, end, ranges)
        check(got, want, name)
        for (x, v), (wx, wy) in zip(got, want):           # the serialised form Qt would receive
            dx, dy = decode(path_bytes(x, v))
            assert np.array_equal(dx, wx) and np.array_equal(dy, wy, equal_nan=True), name
        n += 1
    assert n == 6


@pytest.mark.gpu
def test_gpu_paths_equal_reference_and_oracle(oracle):
    import torch
    from urh_amd.path_creator import create_path_arrays
    for name, samples, start, end, ranges, want in cases():
        check(create_path_arrays(samples, start, end, ranges), want, name)
        if samples.dtype != np.uint16:
            check(create_path_arrays(torch.from_numpy(samples).cuda(), start, end, ranges), want, name + " (device)")
    rng = np.random.default_rng(9)
    for dtype in (np.float32, np.int8, np.uint8, np.int16, np.uint16):
        n = int(rng.integers(20_000, 400_000))
        if dtype == np.float32:
            a = rng.standard_normal(n).astype(np.float32)
            a[rng.integers(0, n, 500)] = np.nan
            a[rng.integers(0, n, 500)] = 0.0
            a[rng.integers(0, n, 500)] = -0.0
            a[rng.integers(0, n, 50)] = np.inf
            a[:n // 4] = np.round(a[:n // 4])                  # many ties incl. +-0
            a[n // 2:n // 2 + 3000] = np.nan                   # whole stretches of NaN
        else:
            info = np.iinfo(dtype)
            a = rng.integers(info.min, info.max + 1, n).astype(dtype)
        for start, end in ((0, n), (17, n - 5), (n // 3, n // 3 + 10_003)):
            got = create_path_arrays(a, start, end)
            want = oracle.create_path_arrays(a, start, end)
            for (x, v), (wx, wv) in zip(got, want):
                assert np.array_equal(x, wx)
                assert v.dtype == wv.dtype and np.array_equal(v.view(np.uint8), wv.view(np.uint8)), (dtype, start, end)
