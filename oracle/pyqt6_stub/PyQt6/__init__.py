"""Minimal PyQt6 stand-in so that the reference's *headless* classes (Signal, ProtocolAnalyzer,
AutoInterpretation, Filter, Modulator) can be imported in this container, where PyQt6 is not
installed.  TEST INFRASTRUCTURE ONLY (used by tests/golden/make_golden.py and oracle/ref_python.py);
written from scratch for this repository -- it contains no reference code."""
