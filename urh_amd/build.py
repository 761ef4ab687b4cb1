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
SOURCES = ["demod_runs.hip", "pulse_table.hip", "filters.hip", "bandpass.hip", "modulate.hip", "plot.hip", "convert.hip", "spectrogram.hip", "costas.hip", "estimators.hip", "msg_estimators.hip", "msg_ranges.hip", "modulation.hip", "compact.hip", "stream.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _newest_source() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hip", ".hpp", ".h")):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib: str = LIB, tag: str = "") -> str:
    """extra_flags / lib / tag: A/B builds of tuning knobs (-DURH_...) into a differently named library that
    URHGPU_LIB=<path> makes urh_amd._lib load (developer tooling; the product is the default build)."""
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _newest_source():
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
