"""Build liburhgpu.so (hand-written HIP for gfx950 + the C ABI of include/urhgpu.h) in-tree.

    python -m urh_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Flags that matter for bit-exactness:
  -ffp-contract=off   the reference's x86-64 build contains no FMA contraction
  (fp32 division / sqrt are hipcc's default correctly rounded forms; fp32 denormals are on).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liburhgpu.so")
SOURCES = ["demod_runs.hip", "pulse_table.hip", "filters.hip", "bandpass.hip", "modulate.hip", "plot.hip", "convert.hip", "fft_peak.hip", "spectrogram.hip", "costas.hip", "estimators.hip", "msg_estimators.hip", "msg_ranges.hip", "modulation.hip", "compact.hip", "stream.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _newest_source() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hip", ".hpp", ".h")):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def _source_hash(extra_flags=()) -> str:
    """sha256 over every source / header of the library and the compiler flags: what a built library is valid for"""
    import hashlib
    h = hashlib.sha256(" ".join([*FLAGS, *extra_flags]).encode())
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".hpp", ".h")):
                h.update(f.encode())
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib: str = LIB, tag: str = "") -> str:
    """extra_flags / lib / tag: A/B builds of tuning knobs (-DURH_...) into a differently named library that
    URHGPU_LIB=<path> makes urh_amd._lib load (developer tooling; the product is the default build).
    A library is reused only when the hash of the sources it was built from (a side file next to it) matches the sources that are there
    now -- an in-tree library of unknown origin (shipped, or older than an edit with a skewed clock) is rebuilt, not trusted."""
    want = _source_hash(extra_flags)
    stamp = lib + ".srchash"
    if not force and os.path.exists(lib) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == want:
                return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        cmd = [hipcc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [src for src, pr in procs if pr.wait() != 0]     # every job is waited for before raising
    if failed:
        raise RuntimeError(f"hipcc failed on {', '.join(failed)}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(want + "\n")
    return lib


if __name__ == "__main__":
    # python -m urh_amd.build [--force] [--tag NAME -DURH_X=1 ...]  (tagged: urh_amd/liburhgpu_NAME.so)
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    flags = [a for a in argv if a.startswith("-D")]
    if tag:
        print(build(force=True, verbose=True, extra_flags=flags, lib=os.path.join(HERE, f"liburhgpu_{tag}.so"), tag="_" + tag))
    else:
        print(build(force="--force" in argv, verbose=True, extra_flags=flags))
