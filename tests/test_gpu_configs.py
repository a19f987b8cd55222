"""-m gpu parity on the workloads of BASELINE configs[3] and configs[4] (one GPU's 4 GiB byte range of each):

  C4  Aho-Corasick, DNA, 8 000 patterns (execute.sh:9,30; main.c:372-373), m = 8 / 16 / 32
  C5  Wu-Manber, 256-symbol alphabet, 100 000 patterns (the reference's maximum, main.c:372), m = 5 / 12 / 20

Each set is checked (i) against the restated search_ac / search_wu2 (oracle/, pinned to the compiled reference on the
golden vectors) on a 32 MiB slice, through both scan engines of the handle, and (ii) on the full 4 GiB shard through
size-independent properties: AC == WM, the sum of 8 byte-range shards (main.c:467-477) == the whole, every engine ==
every other.  Integer work: bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402

pytestmark = pytest.mark.gpu

SLICE = 32 << 20
SHARD = 4 << 30


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if S.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")


class DeviceText:
    """n bytes of the synthetic corpus in HBM (plus 64 readable bytes), generated on the device."""

    def __init__(self, n, sigma):
        self.n = n
        self.ptr = C.c_void_p()
        assert S.lib.smh_device_malloc(C.byref(self.ptr), n + 64) == 0, S.lib.smh_last_error()
        self.cnt = C.c_void_p()
        assert S.lib.smh_device_malloc(C.byref(self.cnt), 16) == 0
        assert S.lib.smh_corpus_text_device(self.ptr, n, 0, 42, sigma, None) == 0
        assert S.lib.smh_stream_synchronize(None) == 0

    def count(self, handle, off, length, variant=S.VARIANT_TUNED):
        assert off % 16 == 0
        S.lib.smh_device_memset(self.cnt, 0, 8, None)
        handle.scan_device(self.ptr.value + off, length, self.cnt.value, variant, None)
        out = np.zeros(1, dtype=np.uint64)
        assert S.lib.smh_copy_to_host(out.ctypes.data_as(C.c_void_p), self.cnt, 8, None) == 0
        assert S.lib.smh_stream_synchronize(None) == 0
        return int(out[0])

    def shards(self, handle, R, m):
        """sum over R byte-range shards with the m-1 halo (main.c:467-477); a shard start that is not 16-byte
        aligned is moved back to the alignment and the columns that adds are counted and subtracted"""
        total = 0
        for r in range(R):
            b, e = S.shard_range(self.n, R, r, m)
            ba = b - (b % 16)
            total += self.count(handle, ba, e - ba)
            if ba != b:
                total -= self.count(handle, ba, b - ba + m - 1)
        return total

    def close(self):
        S.lib.smh_device_free(self.ptr)
        S.lib.smh_device_free(self.cnt)


@pytest.fixture(scope="module")
def dna():
    t = DeviceText(SHARD, 4)
    yield t
    t.close()


@pytest.fixture(scope="module")
def ascii_text():
    t = DeviceText(SHARD, 256)
    yield t
    t.close()


@pytest.mark.parametrize("m", [8, 16, 32])
def test_c4_ac_8000_patterns(dna, m):
    p, sigma = 8000, 4
    pat = S.corpus_patterns(m, p, 10, sigma, 42, SHARD, 2)
    host = S.corpus_text(SLICE, 42, sigma)
    _, tabs = O.oracle_ac(pat, m, p, sigma)
    want = O.oracle_ac_search_tables(host, sigma, tabs)  # search_ac, ac/ac.c:198-222
    assert want > 0
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    engines = {}
    # the Aho-Corasick entry point as it ships (the handle's own engine choice) ...
    engines["ac:" + ("filter" if ac.info().scan_engine == S.ALGO_WM else "automaton")] = ac
    # ... and with the automaton kernels forced (a forced plan always runs them)
    forced = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    forced.set_scan_plan(1, 0)
    assert forced.info().scan_engine == S.ALGO_AC
    engines["ac:automaton(forced)"] = forced
    engines["wm:" + ("automaton" if wm.info().scan_engine == S.ALGO_AC else "filter")] = wm
    if wm.info().scan_engine == S.ALGO_AC:
        own = S.WmTables.from_patterns(pat, m, p, sigma)
        own.set_scan_engine(S.ALGO_WM)
        engines["wm:filter(forced)"] = own
    for name, h in engines.items():
        assert dna.count(h, 0, SLICE) == want, name
    assert dna.count(ac, 0, 1 << 20, S.VARIANT_TABLE) == O.oracle_ac_search_tables(host[:1 << 20], sigma, tabs)
    # the full 4 GiB shard: every engine agrees, and 8 byte-range shards add up to the whole
    whole = dna.count(ac, 0, SHARD)
    assert whole > want
    for name, h in engines.items():
        assert dna.count(h, 0, SHARD) == whole, name
    assert dna.shards(ac, 8, m) == whole
    assert dna.shards(forced, 3, m) == whole


@pytest.mark.parametrize("m", [5, 8, 9, 10, 12, 20])  # 8 / 9: the flat byte-gram form's upper end, 10: the hashed form's first length
def test_c5_wm_100k_patterns_alphabet_256(ascii_text, m):
    p, sigma = 100000, 256
    pat = S.corpus_patterns(m, p, 9, sigma, 42, SHARD, 2)
    host = S.corpus_text(SLICE, 42, sigma)
    csr = O.WMTablesCSR(pat, m, p, sigma)  # preproc_wu2 / search_wu2 (wu/wu.c:151-251) over compressed rows
    want = csr.search(host)
    assert want > 0
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert wm.info().scan_engine == S.ALGO_WM
    assert ascii_text.count(wm, 0, SLICE) == want
    assert ascii_text.count(wm, 0, 1 << 20, S.VARIANT_TABLE) == csr.search(host[:1 << 20])
    whole = ascii_text.count(wm, 0, SHARD)
    assert whole > want
    assert ascii_text.shards(wm, 8, m) == whole
    # the second half of the shard against the oracle as well: the text beyond 2^31 bytes is really scanned
    off = (3 << 30) + (1 << 20)
    tail = S.corpus_text(8 << 20, 42, sigma, offset=off)
    assert ascii_text.count(wm, off, 8 << 20) == csr.search(tail)
