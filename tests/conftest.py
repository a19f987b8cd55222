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


class _Knobs:
    """Development knobs for tests that force a code path (csrc/smh_tune.h).  The product library has none, so such a test
    works on `knob.T`, the wrapper module bound to tests/emu/libsmatcher_hip_testing.so (same sources, -DSMH_TESTING), usually
    as `S = knob.T` in its first line.  The CPU lane emulator (tests/emu/libsmh_emu.so, test harness) mirrors the launchers'
    choices from the environment, so the same string is exported there as well."""
    _ENV = ("SMH_WM_TUNE", "SMH_AC_TUNE", "SMH_HASH_TUNE", "SMH_KEY_TUNE", "SMH_PSET_TUNE")

    def __init__(self, monkeypatch):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-aho-corasick-wu-manber_amd"))
        import smatcher_hip
        self.T = smatcher_hip.load_testing()
        self._mp = monkeypatch

    def set(self, which, text):
        self.T.tune(which, text)
        if text:
            self._mp.setenv(self._ENV[which], text)
        else:
            self._mp.delenv(self._ENV[which], raising=False)

    def wm(self, text):
        self.set(self.T.TUNE_WM, text)

    def ac(self, text):
        self.set(self.T.TUNE_AC, text)

    def pset(self, text):
        self.set(self.T.TUNE_PSET, text)


@pytest.fixture
def knob(monkeypatch):
    k = _Knobs(monkeypatch)
    k.T.tune_clear()
    yield k
    k.T.tune_clear()
