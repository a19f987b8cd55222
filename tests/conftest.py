import os
import sys

import pytest

# Load order matters in processes that use BOTH torch and libsmatcher_hip.so: torch ships its own
# copy of the HIP runtime and loads it by path; libsmatcher_hip.so asks for libamdhip64.so.7 by
# SONAME.  With torch first, the second request resolves to the copy already in the process (one
# runtime).  With libsmatcher_hip.so first, torch adds a SECOND runtime next to /opt/rocm's and
# then sees no GPU ("No HIP GPUs are available").  The tests that mix the two (device-resident
# text in torch tensors, gloo/RCCL sharding) therefore import torch before anything else does.
import torch  # noqa: F401,E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: holds wall-clock / rate expectations (tests/perf.py): recorded under -m gpu, enforced under -m \"gpu and perf\"")
    import perf
    perf.STRICT = "perf" in (config.getoption("markexpr") or "")
