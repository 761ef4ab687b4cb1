#!/usr/bin/env python3
"""Golden vectors for the way OUT of the path (IQArray.tofile / export_to_wav / save_compressed / export_to_sub through
FileOperator.save_data, FileOperator.py:185-196): seeded captures of every sample type are saved with the REAL reference under every
extension; the files' bytes (for `.coco`: the archive member's bytes) are stored in tests/golden/files/export_expected.npz together
with the inputs.  Run in the build container:  python tests/golden/make_export_golden.py"""
import os
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "files")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))

EXTS = [".complex", ".complex16u", ".cu8", ".complex16s", ".cs8", ".complex32u", ".cu16", ".complex32s", ".cs16", ".wav", ".sub", ".coco"]


def inputs():
    rng = np.random.default_rng(424242)
    n = 1200
    env = np.repeat(rng.integers(0, 2, n // 25 + 1), 25)[:n].astype(np.float64)
    clean = np.stack([env * 0.9 - 0.45, np.zeros(n)], axis=1)                       # OOK-like: long runs of two values
    noisy = clean + 0.004 * rng.standard_normal((n, 2))                             # neighbouring uint8 values: the sub writer's one-sample rule
    ph = 2 * np.pi * 0.01 * np.arange(n)
    tone = np.stack([np.cos(ph), np.sin(ph)], axis=1) * 0.8 + 0.05 * rng.standard_normal((n, 2))
    out = {}
    for tag, x in (("clean", clean), ("noisy", noisy), ("tone", tone)):
        out[tag + "_f32"] = x.astype(np.float32)
        out[tag + "_i8"] = np.clip(np.round(x * 127), -128, 127).astype(np.int8)
        out[tag + "_u8"] = np.clip(np.round(x * 127 + 128), 0, 255).astype(np.uint8)
        out[tag + "_i16"] = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)
        out[tag + "_u16"] = np.clip(np.round(x * 32767 + 32768), 0, 65535).astype(np.uint16)
    return out


def main():
    import ref_python
    ref_python.setup()
    from urh.util import FileOperator
    store = {}
    tmp = tempfile.mkdtemp()
    for name, arr in inputs().items():
        store["in/" + name] = arr
        for ext in EXTS:
            for ch in ((1, 2) if ext == ".wav" else (2,)):
                f = os.path.join(tmp, name + ext)
                FileOperator.save_data(arr.copy(), f, sample_rate=250000.0, num_channels=ch)
                if ext == ".coco":
                    with tarfile.open(f, "r") as tar:
                        blob = tar.extractfile(tar.getmembers()[0]).read()
                else:
                    blob = open(f, "rb").read()
                store[f"out/{name}{ext}" + (f"/ch{ch}" if ext == ".wav" else "")] = np.frombuffer(blob, dtype=np.uint8).copy()
                os.remove(f)
    np.savez_compressed(os.path.join(OUT, "export_expected.npz"), **store)
    print(len(store), "entries", os.path.getsize(os.path.join(OUT, "export_expected.npz")), "bytes")


if __name__ == "__main__":
    main()
