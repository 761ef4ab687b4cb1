"""Import the reference's own Python classes (Signal, ProtocolAnalyzer, AutoInterpretation, ...)
on top of the oracle/_ref Cython build.  Where /root/reference exists (this build container) the
sources are read from there; elsewhere (the GPU box) from the copy oracle/build_ref.py staged into
the git-ignored oracle/_ref/pysrc.  Used by tests/golden/make_*.py, by the tests that pin the oracle,
by tests/test_reference_dropin.py and by bench.py's cpu_baseline leg.
TEST INFRASTRUCTURE ONLY."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("URH_REFERENCE", "/root/reference")


def available() -> bool:
    sys.path.insert(0, HERE) if HERE not in sys.path else None
    import build_ref  # noqa
    return build_ref.built() and build_ref.python_src_root() is not None


def setup():
    """Put the PyQt6 stub + reference src on sys.path and graft oracle/_ref's compiled modules
    into the reference's `urh.cythonext` package.  Returns the `urh` package."""
    sys.path.insert(0, HERE)
    import build_ref
    if not build_ref.build():
        raise RuntimeError("reference build unavailable")
    stub = os.path.join(HERE, "pyqt6_stub")
    src = build_ref.python_src_root()
    if src is None:
        raise RuntimeError("reference Python sources unavailable (neither /root/reference nor oracle/_ref/pysrc)")
    for p in (src, stub):
        if p not in sys.path:
            sys.path.insert(0, p)
    # `urh` may already have been imported from oracle/_ref (build_ref.import_ref()); either way make
    # the package span both trees: Python sources from the reference, compiled modules from oracle/_ref.
    import urh
    import urh.cythonext as ce
    for pkg, sub in ((urh, ""), (ce, "cythonext")):
        for base in (os.path.join(src, "urh"), os.path.join(build_ref.OUT, "urh")):
            d = os.path.join(base, sub) if sub else base
            if os.path.isdir(d) and d not in pkg.__path__:
                pkg.__path__.append(d)
    import urh
    return urh
