"""The kernels' lane code (csrc/ac_lane.h, csrc/wm_lane.h), compiled for the CPU and driven thread
by thread over the launch grid, against the reference's golden counts.  This is the GPU-less
check of tiling / halo / tail / early-exit / hot-cold logic; the real kernels are checked through
the C ABI in test_gpu_parity.py (-m gpu)."""
import json
import os

import numpy as np
import pytest

import cases
import emu_lib as E
import oracle_lib as O
from emu_lib import S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
    VECTORS = json.load(f)


def scan_plans(ac, m, sigma):
    """(stride, depth) plans worth exercising for an automaton: exact and cut, both strides."""
    plans = []
    for stride in ((1, 2) if sigma == 4 else (1,)):
        for depth in sorted({m, max(1, m - 1), max(1, m // 2), min(m, 2), min(m, 65)}):
            if depth > 65:
                continue
            try:
                ac.set_scan_plan(stride, depth)
            except S.SmhError:
                continue  # does not fit LDS at this depth / stride
            plans.append((stride, depth))
    if sigma == 4:
        # hybrid stride-2 image (stride code 3): full rows down to depth D, compact item lists below
        for depth in sorted({m, max(4, m - 1), max(4, m // 2)}):
            for full in sorted({1, max(1, depth - 3), max(1, (depth - 3) // 2)}):
                if depth > min(m, 65) or full > depth - 3:
                    continue
                try:
                    ac.set_scan_plan(3, depth | (full << 8))
                except S.SmhError:
                    continue
                plans.append((3, depth | (full << 8)))
    ac.set_scan_plan(0, 0)
    return plans


@pytest.mark.parametrize("vec", VECTORS, ids=[v["name"] for v in VECTORS])
def test_emulated_kernels_match_reference_counts(vec):
    text, pat = cases.build(vec)
    p, m, sigma, want = vec["p"], vec["m"], vec["sigma"], vec["count_ac"]
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    assert E.ac_scan(ac, text, S.VARIANT_TUNED, 3) == want              # the plan the host chose
    assert E.ac_scan(ac, text, S.VARIANT_TABLE, 2) == want              # goto/supply/final walk
    # every scan plan must give the same count: stride 1 / 2, exact (K = m) and depth-cut (K < m)
    for stride, depth in scan_plans(ac, m, sigma):
        ac.set_scan_plan(stride, depth)
        info = ac.info()
        if stride == 3:
            assert info.scan_stride == 2 and info.scan_depth == (depth & 0xFF) and info.scan_full_rows > 0
            assert info.scan_exact == ((depth & 0xFF) == m)
        else:
            assert info.scan_stride == stride and info.scan_depth == depth and info.scan_exact == (depth == m)
            assert info.scan_full_rows == 0
        assert E.ac_scan(ac, text, S.VARIANT_TUNED, 1) == want, (stride, depth)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert E.wm_scan(wm, text, S.VARIANT_TUNED, 3) == want == vec["count_wu2"]
    assert E.wm_scan(wm, text, S.VARIANT_TABLE, 2) == want


def test_every_boundary_offset():
    """A single pattern occurrence slid across segment (64 B), wave-chunk (4/8 KiB) and text-end
    boundaries; the count must be exactly 1 (or 2 when planted twice) at every offset."""
    m, sigma = 8, 4
    pat = np.array([3, 1, 2, 0, 3, 3, 1, 2], dtype=np.uint8)
    ac = S.AcAutomaton.from_patterns(pat, m, 1, sigma)
    wm = S.WmTables.from_patterns(pat, m, 1, sigma)
    n = 8192 * 2 + 64 + 16 + 5
    offsets = list(range(0, 80)) + list(range(4096 - 70, 4096 + 10)) + list(range(8192 - 70, 8192 + 30)) + \
        list(range(16384 - 10, n - m + 1))
    for off in offsets:
        text = np.zeros(n, dtype=np.uint8)
        text[off:off + m] = pat
        got = [E.ac_scan(ac, text, 1, 1), E.wm_scan(wm, text, 0, 1), E.wm_scan(wm, text, 1, 1)]
        for stride, depth in ((1, 8), (2, 8), (1, 5), (2, 5), (2, 4), (1, 1), (3, 8 | (2 << 8)), (3, 7 | (4 << 8)),
                              (3, 5 | (1 << 8))):
            ac.set_scan_plan(stride, depth)
            got.append(E.ac_scan(ac, text, 0, 1))
        assert got == [1] * len(got), (off, got)


@pytest.mark.parametrize("m", [9, 24, 40, 70])
def test_long_patterns_straddling_segments(m):
    sigma = 4
    rng = np.random.RandomState(m)
    pat = rng.randint(0, sigma, size=m).astype(np.uint8)
    ac = S.AcAutomaton.from_patterns(pat, m, 1, sigma)
    wm = S.WmTables.from_patterns(pat, m, 1, sigma)
    n = 8192 + 4096 + 200
    for off in list(range(0, 70, 3)) + list(range(8192 - m - 2, 8192 + 3)) + [n - m]:
        text = rng.randint(0, sigma, size=n).astype(np.uint8)
        text[off:off + m] = pat
        want = O.count_bruteforce(pat, m, 1, text)
        assert want >= 1
        for stride, depth in ((0, 0), (1, min(m, 65)), (2, min(m, 65)), (1, 7), (2, 7), (2, 6), (1, 3),
                              (3, min(m, 65) | (1 << 8)), (3, min(m, 65) | (5 << 8)), (3, min(m, 33) | (4 << 8)),
                              (3, 9 | (3 << 8)), (3, 6 | (2 << 8))):
            if stride == 3 and (depth & 0xFF) < m and (depth & 0xFF) > 33:
                continue  # a depth-cut hybrid image records candidates as bits: halo K - 1 <= 32
            ac.set_scan_plan(stride, depth)
            assert E.ac_scan(ac, text, 0, 1) == want, (stride, depth)
        assert E.wm_scan(wm, text, 0, 1) == want
        assert E.wm_scan(wm, text, 1, 1) == want


def test_grid_size_does_not_change_the_count():
    vec = next(v for v in VECTORS if v["name"] == "dense_dna")
    text, pat = cases.build(vec)
    ac = S.AcAutomaton.from_patterns(pat, vec["m"], vec["p"], vec["sigma"])
    wm = S.WmTables.from_patterns(pat, vec["m"], vec["p"], vec["sigma"])
    for blocks in (1, 2, 5, 16):
        assert E.ac_scan(ac, text, 0, blocks) == vec["count_ac"]
        assert E.wm_scan(wm, text, 0, blocks) == vec["count_ac"]


@pytest.mark.parametrize("kind,sigma,m,p", [(1, 4, 11, 40), (1, 4, 16, 300), (1, 4, 14, 4000), (1, 4, 21, 9000), (1, 4, 33, 50), (3, 4, 11, 200), (3, 4, 16, 2000),
                                            (6, 256, 5, 300), (6, 256, 6, 3000), (6, 256, 7, 100), (6, 256, 8, 5000), (6, 128, 7, 100), (6, 256, 12, 3000), (6, 256, 20, 500),
                                            (5, 4, 11, 30), (5, 4, 12, 200), (5, 4, 16, 8000), (5, 4, 17, 500), (5, 4, 18, 3000), (5, 4, 23, 20000), (5, 4, 24, 100), (5, 4, 33, 50),
                                            (3, 4, 32, 500), (2, 256, 5, 300), (2, 256, 12, 3000), (2, 256, 20, 500),
                                            (2, 128, 7, 100),
                                            (8, 256, 5, 300), (8, 256, 12, 3000), (8, 256, 17, 2000), (8, 256, 20, 500), (8, 256, 33, 200), (8, 20, 10, 500),
                                            (9, 256, 5, 300), (9, 256, 6, 3000), (9, 256, 8, 5000), (9, 256, 12, 3000), (9, 256, 20, 500), (9, 128, 7, 100),
                                            (11, 20, 8, 10000), (11, 20, 6, 300), (11, 20, 12, 3000), (11, 256, 7, 3000), (11, 256, 20, 500), (11, 128, 33, 100)])
def test_gram_filter_forms(kind, sigma, m, p, knob):
    """The three q-gram shift-or forms (symbol pairs, 8-symbol grams, hashed byte grams), each forced with the
    development knob so the test does not depend on the cost model: random text with planted occurrences,
    including ones that straddle segment / wave-chunk boundaries and the text's first and last columns."""
    S = knob.T  # the testing build: the knobs below exist only there
    knob.wm("gram=%d" % kind)
    rng = np.random.RandomState(1000 * kind + m)
    n = 3 * 4096 + 777
    text = rng.randint(0, sigma, size=n).astype(np.uint8)
    pat = rng.randint(0, sigma, size=(p, m)).astype(np.uint8)
    for i, off in enumerate([0, 300, 640 - m // 2, 4096 - m // 2, 8191, 8192 + 2 * m + 64, n - m]):
        if 0 <= off <= n - m:
            text[off:off + m] = pat[(7 * i) % p]
    pat[p // 2] = pat[0]  # a duplicate pattern: a column is counted once
    wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, sigma)
    info = wm.info()
    assert info.gram_planes == min({1: 15, 5: 16}.get(kind, 8), m - {1: 6, 3: 7, 2: 2, 5: 7, 6: 2, 8: 2, 9: 2, 11: 3}[kind]) and info.gram_kind == kind
    assert info.lds_bytes == {3: 65536, 8: 147392, 9: 147392, 11: 147392}.get(kind, 131072)
    if info.scan_engine != S.ALGO_WM:
        wm.set_scan_engine(S.ALGO_WM)
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    assert want >= 7
    for blocks in (1, 3):
        assert E.wm_scan(wm, text, S.VARIANT_TUNED, blocks) == want
    total, pos = E.wm_positions(wm, text, want + 8, 2)
    assert total == want and len(set(pos.tolist())) == want



@pytest.mark.parametrize("kind,m,p", [(6, 5, 300), (6, 8, 5000), (6, 17, 400), (6, 20, 500), (2, 5, 300), (2, 12, 3000), (2, 17, 300), (2, 33, 200)])
def test_byte_gram_forms_staged_and_from_l2(kind, m, p, knob):
    """The byte forms' two verify modes (round 4: windows from L2, the default; SMH_WM_TUNE="l2=0": the chunk staged in LDS)
    give the count of the definition; the window request bends dwords the window does not reach back onto its last one."""
    S = knob.T  # the testing build: the knobs below exist only there
    rng = np.random.RandomState(31 * kind + m)
    n = 3 * 4096 + 999
    text = rng.randint(0, 256, size=n).astype(np.uint8)
    pat = rng.randint(0, 256, size=(p, m)).astype(np.uint8)
    for i, off in enumerate([0, 300, 640 - m // 2, 4096 - m // 2, 8191, 8192 + 2 * m + 64, 2 * 4096 - 1, 2 * 4096, 3 * 4096 - m, n - m]):
        text[off:off + m] = pat[(7 * i) % p]
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    for tune in ("gram=%d" % kind, "gram=%d,l2=0" % kind):
        knob.wm(tune)
        wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, 256)
        assert wm.info().gram_kind == kind
        for blocks in (1, 3):
            assert E.wm_scan(wm, text, S.VARIANT_TUNED, blocks) == want, tune
        total, pos = E.wm_positions(wm, text, want + 8, 2)
        assert total == want and len(set(pos.tolist())) == want


@pytest.mark.parametrize("form", [6, 9])
@pytest.mark.parametrize("m,p", [(5, 300), (5, 20000), (6, 3000), (7, 100), (7, 9000)])
def test_flat_byte_grams_with_two_bits_per_gram(m, p, form, knob):
    """Round 4: patterns of 5..7 bytes may keep TWO bits per gram in the flat Bloom set (a blocked Bloom filter with 8-bit
    blocks: wm_lane.h smh_flat_addr<true>); forced here with the development knob, against brute force, both block counts,
    positions mode, and the bounds-checked first / last chunks."""
    S = knob.T  # the testing build: the knobs below exist only there
    knob.wm("gram=%d,flatk=2" % form)  # 9: the same set in the 143.9 KiB table (round 6)
    rng = np.random.RandomState(77 * m + p)
    n = 3 * 4096 + 555
    text = rng.randint(0, 256, size=n).astype(np.uint8)
    pat = rng.randint(0, 256, size=(p, m)).astype(np.uint8)
    for i, off in enumerate([0, 300, 640 - m // 2, 4096 - m // 2, 8191, 8192 + 2 * m + 64, n - m]):
        text[off:off + m] = pat[(7 * i) % p]
    wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, 256)
    assert wm.info().gram_kind == form
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    assert want >= 7
    for blocks in (1, 3):
        assert E.wm_scan(wm, text, S.VARIANT_TUNED, blocks) == want
    total, pos = E.wm_positions(wm, text, want + 8, 2)
    assert total == want and len(set(pos.tolist())) == want
    knob.wm("gram=%d,flatk=1" % form)
    one = S.WmTables.from_patterns(pat.reshape(-1), m, p, 256)
    assert E.wm_scan(one, text, S.VARIANT_TUNED, 2) == want


@pytest.mark.parametrize("kind", [1, 5])
@pytest.mark.parametrize("m", [11, 12, 13, 14, 16, 17, 18, 21, 24, 29, 32, 33])
def test_in_register_verify_every_column_and_length(m, kind, knob):
    """Round 3: the pair form's in-register verify (wm_lane.h smh_regv_tag: the window's dwords selected out of the lane's
    text registers and the previous lane's tail by a barrel of conditional moves).  One occurrence ending at EVERY column
    residue 0..63 of a segment -- in lane 0 of a wave-chunk (window out of the halo), in lanes 1 and 63, across a chunk
    boundary -- for every window length class (whole dwords, +1, +2, +3 bytes; 16 and 32 bytes of halo); forced on, forced off
    (staged verify), both equal to the definition."""
    S = knob.T  # the testing build: the knobs below exist only there
    sigma, p = 4, 64 * 4
    rng = np.random.RandomState(500 + m)
    n = 4 * 4096 + 100
    text = rng.randint(0, sigma, size=n).astype(np.uint8)
    ends = []
    for r in range(64):
        ends += [4096 + r,                     # lane 0 of chunk 1: the window reaches into the bytes in front of the chunk
                 4096 + 64 * (1 + r % 3) + r,  # lanes 1..3
                 2 * 4096 + 64 * 63 + r,       # lane 63
                 3 * 4096 + 64 * (r % 64) + r] # every lane once
    pat = np.stack([text[e - m + 1:e + 1] for e in ends]).astype(np.uint8)
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    assert want >= len(set(ends))
    for tune in ("gram=%d,regv=1" % kind, "gram=%d,regv=0" % kind, "gram=%d,l2=1" % kind):  # (l2=1, late round 6: the windows-from-L2 pipeline)
        knob.wm(tune)
        wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, sigma)
        if wm.info().scan_engine != S.ALGO_WM:
            wm.set_scan_engine(S.ALGO_WM)
        assert wm.info().gram_planes == (min(15, m - 6) if kind == 1 else min(16, m - 7))
        for blocks in (1, 2):
            assert E.wm_scan(wm, text, S.VARIANT_TUNED, blocks) == want, tune
        total, pos = E.wm_positions(wm, text, want + 8, 2)
        assert total == want and set(ends) <= set(int(x) for x in pos), tune


@pytest.mark.parametrize("m,p", [(8, 8000), (8, 20000), (6, 3000), (3, 40), (5, 900), (8, 3)])
def test_dense_plan(m, p):
    """The dense plan of the automaton engine (alphabet 4, m <= 8: state = the last m symbols, acceptance one bit per
    string, ac_host.c dense_build): chosen by itself where the stride-2 image does not fit, forced (stride code 4) on the
    others; the accepting bits come from walking the compiled DFA, so duplicates and dense sets are the interesting cases."""
    sigma, n = 4, 70_001
    rng = np.random.RandomState(m * 1000 + p)
    pat = rng.randint(0, sigma, size=m * p).astype(np.uint8)
    pat[:m] = pat[m:2 * m] if p > 1 else pat[:m]  # a duplicate
    text = rng.randint(0, sigma, size=n).astype(np.uint8)
    text[:m] = pat[:m]
    text[n - m:] = pat[-m:]
    want = O.count_bruteforce(pat, m, p, text)
    assert want >= 2
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    chosen = ac.info().scan_dense
    assert chosen == (1 if (m, p) in ((8, 8000), (8, 20000)) else chosen)  # the big 8-mer sets take it by themselves
    assert E.ac_scan(ac, text, 0, 2) == want
    ac.set_scan_plan(4, 0)
    assert ac.info().scan_dense == 1 and ac.info().scan_engine == S.ALGO_AC
    for blocks in (1, 3):
        assert E.ac_scan(ac, text, 0, blocks) == want
    ac.set_scan_plan(1, 0)
    assert ac.info().scan_dense == 0 and E.ac_scan(ac, text, 0, 1) == want
    ac.set_scan_plan(0, 0)
    assert ac.info().scan_dense == chosen
    with pytest.raises(S.SmhError):
        S.AcAutomaton.from_patterns(rng.randint(0, sigma, size=9 * 10).astype(np.uint8), 9, 10, sigma).set_scan_plan(4, 0)


def test_cuckoo_form_of_the_verify_table():
    """sets whose four-slots-per-pattern bucket table exceeds 512 KiB also keep their verify entries as a two-table cuckoo hash
    (82 % full), which the pipelined probes read (csrc/wm_lane.h smh_wm_ck_*): same counts as the bucket table's, on uniform text
    and on text made of the patterns themselves"""
    import smatcher_hip as S
    sigma, p = 256, 40000
    for m in (8, 12, 20):
        pat = O.gen_patterns(m, p, 7, sigma)
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        info = wm.info()
        assert info.verify_ck_slots >= info.distinct and info.verify_ck_slots * 4 < info.verify_slots * 4 / 2, (info.verify_ck_slots, info.verify_slots)
        wm.set_scan_engine(S.ALGO_WM)
        text = np.concatenate([O.gen_text(3 * 4096 + 100, 9, sigma), pat[:m * 3000], O.gen_text(5000, 10, sigma)])
        want = O.count_bruteforce(pat, m, p, text)
        assert want >= 3000
        assert E.wm_scan(wm, text) == want
        wm.close()
    small = S.WmTables.from_patterns(O.gen_patterns(8, 3000, 7, 256), 8, 3000, 256)
    assert small.info().verify_ck_slots == 0
    small.close()
