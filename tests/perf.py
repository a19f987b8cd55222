"""Wall-clock / rate / "within N launches" expectations of the GPU tests.

A parity run (`pytest -m gpu`) must not turn red because a box is noisy: there, perf_check() only RECORDS a missed
expectation (a pytest warning, and a line in gpurun_out/perf_notes.log when that directory exists).  The tests that hold such
expectations also carry the `perf` marker; `pytest -m "gpu and perf"` runs exactly those with the expectations enforced
(conftest.py sets STRICT from the -m expression).  The count / parity assertions of those tests are plain asserts in both runs."""
import os
import warnings

STRICT = False
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def perf_check(ok, what):
    """`what` = a sentence with the measured number in it"""
    if ok:
        return True
    if STRICT:
        raise AssertionError("perf expectation missed: " + what)
    warnings.warn("perf expectation missed (not enforced without -m perf): " + what)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "perf_notes.log"), "a") as f:
            f.write(what + "\n")
    return False
