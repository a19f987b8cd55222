"""SOG (third sibling): preproc_sog8 / search_sog8 / cuda_sog1..5 (smatcher.h:108-109, sog/sog8.c,
cuda/cuda_sog.cu).

tests/golden/ref_sog_vectors.json was produced by RUNNING THE REFERENCE (make_golden_sog.py): digests of the tables
its preproc_sog8 fills deterministically (T8, sorted hashes, permutation) and the count of 8-byte windows that equal a
pattern (the reference's search_ac on the same input).  The reference's own search_sog8 count is NOT the oracle: its
2-level bitmap comes from an uninitialised variable (sog/sog8.c:124,135) -- `count_ref_sog8` in the file shows it
dropping matches on 5 of the 8 cases.  CPU: the restatement (oracle/ora_sog.c), the library's preproc_sog8 and the
emulated table-walking lane code reproduce tables and counts.  GPU: search_sog8 and both kernel families."""
import json
import os

import numpy as np
import pytest

import cases
import emu_lib as E
import oracle_lib as O
from emu_lib import S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "ref_sog_vectors.json")) as f:
    VECTORS = json.load(f)
IDS = [v["name"] for v in VECTORS]


def hx(a):
    return "%016x" % O.fnv(a)


@pytest.mark.parametrize("vec", VECTORS, ids=IDS)
def test_tables_and_counts_match_the_reference(vec):
    text, pat = cases.build(vec)
    p = vec["p"]
    cnt, t = O.oracle_sog8(pat, p, text)
    assert (hx(t.T8), hx(t.scanner_hs), hx(t.scanner_index)) == (vec["fnv_T8"], vec["fnv_hs"], vec["fnv_index"])
    assert cnt == vec["count"] == O.count_bruteforce(pat, 8, p, text)
    # the library's host code fills the caller's tables the same way, the 2-level bitmap with the defined contents
    sg = S.SogTables(pat, p)
    assert (hx(sg.T8), hx(sg.scanner_hs), hx(sg.scanner_index)) == (vec["fnv_T8"], vec["fnv_hs"], vec["fnv_index"])
    assert np.array_equal(sg.scanner_hs2, t.scanner_hs2)
    # the table-walking lane code over those tables (CPU emulation)
    assert E.sog_scan(sg, text, 2) == vec["count"]
    assert E.sog_scan(sg, text[:7], 1) == 0
    sg.close()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libref.so not built (no /root/reference here)")
@pytest.mark.parametrize("seed", range(4))
def test_tables_against_live_reference(seed):
    rng = np.random.RandomState(seed)
    sigma = [4, 256, 20, 2][seed]
    p = int(rng.randint(1, 700))
    n = int(rng.randint(8, 60000))
    text = O.gen_text(n, 3000 + seed, sigma)
    pat = O.gen_patterns_mixed(8, p, 4000 + seed, sigma, 3000 + seed, n, 3)
    ref_t = O.SogTables(p)
    O.ref_sog8(pat, p, text, ref_t)
    cnt, t = O.oracle_sog8(pat, p, text)
    assert np.array_equal(t.T8, ref_t.T8) and np.array_equal(t.scanner_hs, ref_t.scanner_hs)
    assert np.array_equal(t.scanner_index, ref_t.scanner_index)
    assert cnt == O.ref_ac(pat, 8, p, sigma, text)[0]


def test_sog_needs_a_gpu_to_search():
    if S.device_count() > 0:
        pytest.skip("a device is visible")
    sg = S.SogTables(O.gen_patterns(8, 5, 1, 4), 5)
    with pytest.raises(S.SmhError):
        sg.count_host(np.zeros(100, dtype=np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("vec", VECTORS, ids=IDS)
def test_gpu_sog_counts(vec):
    text, pat = cases.build(vec)
    p, want = vec["p"], vec["count"]
    sg = S.SogTables(pat, p)
    assert sg.count_host(text, S.VARIANT_TUNED)[0] == want
    assert sg.count_host(text, S.VARIANT_TABLE)[0] == want
    # the legacy entry point with the reference's argument list
    txt, tp = S._u8(text)
    got = S.lib.search_sog8(*sg.tables(), sg.ptrs, 8, tp, len(txt), p, 3)
    assert got == want
    sg.close()
