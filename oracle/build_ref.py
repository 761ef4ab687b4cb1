#!/usr/bin/env python3
"""Build the REAL reference (jopohl/urh Cython extensions) into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.

The reference's hot path lives in three Cython modules
(/root/reference/src/urh/cythonext/{signal_functions,util,auto_interpretation}.pyx).
This recipe cythonizes them *from where they lie* (no reference source is copied
into the repository) with exactly the reference's own settings:

  * compiler directives   -> /root/reference/src/urh/dev/native/ExtensionHelper.py:14-20
  * language="c++", -fopenmp -> /root/reference/setup.py:23-30, 108-113

Generated .cpp files and the built .so files go to oracle/_ref/ only
(git-ignored, NOT gpurun-ignored: the .so files travel to the GPU box, where
/root/reference does not exist).  When /root/reference is absent this script is a
no-op and whatever is already in oracle/_ref/ is used.
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("URH_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")
PKG = os.path.join(OUT, "urh", "cythonext")
MODULES = ["util", "signal_functions", "auto_interpretation", "path_creator"]   # path_creator imports only under ref_python.setup() (PyQt6 stub)

DIRECTIVES = {
    "language_level": 3,
    "cdivision": True,
    "wraparound": False,
    "boundscheck": False,
    "initializedcheck": False,
}


def ref_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "urh", "cythonext"))


def built() -> bool:
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return all(os.path.exists(os.path.join(PKG, m + suffix)) for m in MODULES)


def build(force: bool = False) -> bool:
    """Returns True when oracle/_ref holds a usable build."""
    if built() and not force:
        return True
    if not ref_available():
        return built()
    import numpy as np
    from Cython.Build import cythonize  # noqa: F401  (availability check)
    from Cython.Compiler import Options  # noqa: F401

    src_dir = os.path.join(REF_ROOT, "src", "urh", "cythonext")
    gen = os.path.join(OUT, "gen", "urh", "cythonext")
    os.makedirs(gen, exist_ok=True)
    os.makedirs(PKG, exist_ok=True)
    for d in (os.path.join(OUT, "urh"), PKG):
        init = os.path.join(d, "__init__.py")
        if not os.path.exists(init):
            open(init, "w").close()

    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    py_inc = sysconfig.get_paths()["include"]
    np_inc = np.get_include()
    dir_args = []
    for k, v in DIRECTIVES.items():
        dir_args += ["-X", f"{k}={v}"]
    for m in MODULES:
        pyx = os.path.join(src_dir, m + ".pyx")
        cpp = os.path.join(gen, m + ".cpp")
        # cython resolves `from urh.cythonext.util cimport ...` through -I <src>
        cmd = [sys.executable, "-m", "cython", "--cplus", "-3",
               "-I", os.path.join(REF_ROOT, "src"), *dir_args, "-o", cpp, pyx]
        subprocess.check_call(cmd, cwd=OUT)
        so = os.path.join(PKG, m + suffix)
        cc = ["g++", "-O2", "-fPIC", "-shared", "-fopenmp", "-w",
              "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
              "-I", py_inc, "-I", np_inc, cpp, "-o", so]
        subprocess.check_call(cc)
    shutil.rmtree(os.path.join(OUT, "gen"), ignore_errors=True)
    return built()


# Python side of the reference (Signal, ProtocolAnalyzer, AutoInterpretation, ...) and the captures its own demodulation tests
# read: STAGED into the git-ignored oracle/_ref/ (never into history) so that the GPU box -- where /root/reference does not
# exist -- can run the reference's classes with this library patched in underneath (tests/test_reference_dropin.py) and
# bench.py can time the reference's own end-to-end Python path as the CPU baseline.
PYSRC = os.path.join(OUT, "pysrc")
REFTESTS = os.path.join(OUT, "reftests")
# the reference's headless hot-path tests (SURVEY.md probe table): staged next to the captures they read, with ONE textual change
# made on the staged copy: `from tests.test_util import get_path_for_data_file` -> `from tests.utils_testing import ...`
# (tests/test_util.py:8 imports the GUI QtTestCase, which needs a display; utils_testing.py holds the function itself)
STAGED_TESTS = ["test_demodulations.py", "utils_testing.py", "__init__.py", "test_protocol_analyzer.py", "test_iq_array.py",
                "test_modulator.py"]
STAGED_TEST_DIRS = ["auto_interpretation"]


def staged() -> bool:
    return os.path.exists(os.path.join(PYSRC, "urh", "signalprocessing", "Signal.py")) and \
        os.path.exists(os.path.join(REFTESTS, "tests", "test_demodulations.py")) and \
        os.path.exists(os.path.join(REFTESTS, "tests", "auto_interpretation", "test_center_detection.py")) and \
        os.path.exists(os.path.join(REFTESTS, "tests", "data", "xavax.coco"))


def _stage_test_file(src: str, dst: str):
    with open(src, "r", encoding="utf-8") as f:
        text = f.read()
    text = text.replace("from tests.test_util import get_path_for_data_file", "from tests.utils_testing import get_path_for_data_file")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w", encoding="utf-8") as f:
        f.write(text)


def stage_python(force: bool = False) -> bool:
    if staged() and not force:
        return True
    if not ref_available():
        return staged()
    src = os.path.join(REF_ROOT, "src", "urh")
    for root, dirs, files in os.walk(src):
        dirs[:] = [d for d in dirs if d != "__pycache__"]
        rel = os.path.relpath(root, src)
        for f in files:
            if f.endswith(".py"):
                dst = os.path.join(PYSRC, "urh", rel, f)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copyfile(os.path.join(root, f), dst)
    tdst = os.path.join(REFTESTS, "tests")
    os.makedirs(os.path.join(tdst, "data"), exist_ok=True)
    for f in STAGED_TESTS:
        a = os.path.join(REF_ROOT, "tests", f)
        if os.path.exists(a):
            _stage_test_file(a, os.path.join(tdst, f))
    for d in STAGED_TEST_DIRS:
        for f in sorted(os.listdir(os.path.join(REF_ROOT, "tests", d))):
            if f.endswith(".py"):
                _stage_test_file(os.path.join(REF_ROOT, "tests", d, f), os.path.join(tdst, d, f))
    ddir = os.path.join(REF_ROOT, "tests", "data")
    for f in sorted(os.listdir(ddir)):                     # every capture the staged tests read (18 MB in all)
        a = os.path.join(ddir, f)
        if os.path.isfile(a):
            shutil.copyfile(a, os.path.join(tdst, "data", f))
    return staged()


def python_src_root():
    """Directory holding the reference's `urh` Python package: the reference tree itself where it exists, else the staged copy."""
    if ref_available():
        return os.path.join(REF_ROOT, "src")
    return PYSRC if staged() else None


def import_ref():
    """Import the built reference modules (signal_functions, util, auto_interpretation)."""
    if not built():
        raise RuntimeError("oracle/_ref is not built (run oracle/build_ref.py where /root/reference exists)")
    if OUT not in sys.path:
        sys.path.insert(0, OUT)
    from urh.cythonext import signal_functions, util, auto_interpretation  # type: ignore
    return signal_functions, util, auto_interpretation


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref built:", ok, " python sources staged:", stage_python(force="--force" in sys.argv))
    sys.exit(0 if ok else 1)
