"""The oracle (oracle/ora_*.c) against the reference: committed golden vectors produced by the
reference's own compiled ac/ac.c + wu/wu.c (tests/golden/make_golden.py), SURVEY.md 8c known
answers, and -- where oracle/_ref/libref.so is present -- the live reference on fresh seeds."""
import json
import os

import numpy as np
import pytest

import cases
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "ref_vectors.json")) as f:
    VECTORS = json.load(f)
BY_NAME = {v["name"]: v for v in VECTORS}


def hx(x):
    return "%016x" % x


@pytest.mark.parametrize("vec", VECTORS, ids=[v["name"] for v in VECTORS])
def test_oracle_reproduces_reference_vectors(vec):
    text, pat = cases.build(vec)
    n, p, m, sigma = vec["n"], vec["p"], vec["m"], vec["sigma"]
    c_ac, t = O.oracle_ac(pat, m, p, sigma, text)
    assert c_ac == vec["count_ac"]
    assert t.idcounter == vec["idcounter"] and t.patterncounter == vec["patterncounter"]
    assert hx(O.fnv(t.state_transition[:t.idcounter * sigma])) == vec["fnv_transition"]
    assert hx(O.fnv(t.state_supply[:t.idcounter])) == vec["fnv_supply"]
    assert hx(O.fnv(t.state_final[:t.idcounter])) == vec["fnv_final"]
    for flat, key in ((True, "count_wu2"), (False, "count_wu")):
        c_wu, tw = O.oracle_wu(pat, m, p, sigma, text, flat=flat)
        assert c_wu == vec[key]
        assert [hx(d) for d in tw.digest()] == vec["fnv_wm"]
    # the compressed-row form (what checks alphabet 256 / 100 000 patterns): same tables, same count
    csr = O.WMTablesCSR(pat, m, p, sigma)
    assert [hx(d) for d in csr.digest()] == vec["fnv_wm"]
    assert csr.search(text) == vec["count_wu2"]
    # the reference's two algorithms agree with each other and with the definition
    assert vec["count_ac"] == vec["count_wu"] == vec["count_wu2"]
    if n <= 70000:
        assert O.count_bruteforce(pat, m, p, text) == c_ac


def test_survey_known_answers():
    """counts measured on the compiled reference during the survey."""
    assert BY_NAME["kat_1m_100x8"]["count_ac"] == 1539
    assert BY_NAME["kat_1m_100x8"]["idcounter"] == 549
    text = O.gen_text(1 << 20, 42, 4)
    pat = O.gen_patterns(8, 100, 7, 4)
    c, t = O.oracle_ac(pat, 8, 100, 4, text)
    assert (c, t.idcounter, t.patterncounter) == (1539, 549, 100)
    # 8-way shard of the same case with the main.c:467-477 formula
    per_rank = []
    for i in range(8):
        b, e = O.shard_range(1 << 20, 8, i, 8)
        per_rank.append(O.oracle_ac(pat, 8, 100, 4, text[b:e])[0])
    assert per_rank == [215, 208, 203, 195, 202, 163, 205, 148] and sum(per_rank) == 1539
    text3 = O.gen_text(1000003, 42, 4)
    tot = 0
    for i in range(3):
        b, e = O.shard_range(1000003, 3, i, 8)
        tot += O.oracle_wu(pat, 8, 100, 4, text3[b:e])[0]
    assert tot == 1483 == BY_NAME["kat_1m003_r3"]["count_ac"]


def test_hand_kat_tables():
    """SURVEY.md 8c hand KAT: sigma 4, m 4, patterns 0123, 1230, 0123 (duplicate)."""
    pat = np.array([0, 1, 2, 3, 1, 2, 3, 0, 0, 1, 2, 3], dtype=np.uint8)
    text = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 0, 1, 2, 3, 3, 3, 1, 2, 3, 0, 1], dtype=np.uint8)
    c, t = O.oracle_ac(pat, 4, 3, 4, text)
    assert c == 6 and t.idcounter == 9 and t.patterncounter == 2
    rows = t.state_transition[:9 * 4].reshape(9, 4).tolist()
    assert rows == [[1, 5, 0, 0], [-1, 2, -1, -1], [-1, -1, 3, -1], [-1, -1, -1, 4], [-1] * 4,
                    [-1, -1, 6, -1], [-1, -1, -1, 7], [8, -1, -1, -1], [-1] * 4]
    assert t.state_supply[:9].tolist() == [0, 0, 5, 6, 7, 0, 0, 0, 1]
    assert np.nonzero(t.state_final[:9])[0].tolist() == [4, 8]
    assert O.oracle_wu(pat, 4, 3, 4, text)[0] == 6
    assert O.positions_bruteforce(pat, 4, 3, text).tolist() == [3, 4, 7, 8, 12, 18]


def test_shiftsize_table():
    # wu/wu.c:18-47
    want = {2: 22, 4: 64, 8: 148, 20: 400, 128: 2668, 256: 5356, 512: 10732, 1024: 21484}
    for a, s in want.items():
        assert O.lib.ora_wu_determine_shiftsize(a) == s
    assert O.lib.ora_wu_determine_shiftsize(5) == 0


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libref.so not built (no /root/reference here)")
@pytest.mark.parametrize("seed", range(12))
def test_oracle_against_live_reference(seed):
    rng = np.random.RandomState(seed)
    sigma = [2, 4, 8, 20, 128, 256][seed % 6]
    m = int(rng.randint(3, 40))
    p = int(rng.randint(1, 400))
    n = int(rng.randint(m, 50000))
    text = O.gen_text(n, 1000 + seed, sigma)
    pat = O.gen_patterns_mixed(m, p, 2000 + seed, sigma, 1000 + seed, n, 3)
    c_r, t_r, _, _ = O.ref_ac(pat, m, p, sigma, text)
    c_o, t_o = O.oracle_ac(pat, m, p, sigma, text)
    assert c_o == c_r and t_o.idcounter == t_r.idcounter and t_o.patterncounter == t_r.patterncounter
    assert np.array_equal(t_o.state_transition, t_r.state_transition)
    assert np.array_equal(t_o.state_supply, t_r.state_supply)
    assert np.array_equal(t_o.state_final, t_r.state_final)
    for flat in (True, False):
        cw_r, tw_r, _, _ = O.ref_wu(pat, m, p, sigma, text, flat=flat)
        cw_o, tw_o = O.oracle_wu(pat, m, p, sigma, text, flat=flat)
        assert cw_o == cw_r == c_r
        assert np.array_equal(tw_o.SHIFT, tw_r.SHIFT) and np.array_equal(tw_o.PREFIX_size, tw_r.PREFIX_size)
        assert tw_o.digest() == tw_r.digest()
    csr = O.WMTablesCSR(pat, m, p, sigma)
    assert csr.digest() == tw_r.digest() and csr.search(text) == c_r
