"""Pattern sets with mixed lengths (include/smatcher_hip.h smh_pset_*).

The reference takes one pattern length per run, so the expected value of a mixed set is the
length-class decomposition -- tests/golden/ref_mixed_vectors.json holds, per class, the counts the
reference's own search_ac / search_wu2 produced (tests/golden/make_golden_mixed.py), and their sum.
CPU: the oracle and the emulated lane code reproduce every class count (including the length-1 and
length-2 classes Wu-Manber cannot take); the set handle groups classes correctly and rejects bad
input.  GPU: smh_pset_count_host / smh_pset_scan / smh_pset_positions on the device equal the totals,
for both algorithms."""
import json
import os

import numpy as np
import pytest

import cases
import emu_lib as E
import oracle_lib as O
from emu_lib import S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "ref_mixed_vectors.json")) as f:
    MIXED = json.load(f)
IDS = [v["name"] for v in MIXED]
# SMH_ALGO_WM sets whose min-length suffix filter passes < 0.4 % of the columns: one pass with that filter (the
# others take the grouped pair-gram filter when their short patterns are few, else the automaton of ONE_PASS_AC)
ONE_PASS = {"mixed_ascii_5_20"}
# SMH_ALGO_AC sets: one automaton with joined output counts whenever a cut of it fits LDS with few candidates
# (alphabet 256: a row costs 512 bytes, only depth 1 fits, every position would be a candidate: per class)
ONE_PASS_AC = {"mixed_dna_8_32", "mixed_protein", "mixed_len_1_2", "mixed_prefix_hazard"}


@pytest.mark.parametrize("vec", MIXED, ids=IDS)
def test_oracle_and_lane_code_reproduce_every_class(vec):
    text, patterns, lengths = cases.build_mixed(vec)
    classes = cases.split_classes(patterns, lengths)
    assert sorted(classes) == [c["length"] for c in vec["per_class"]]
    total = 0
    for c in vec["per_class"]:
        L, p, sigma = c["length"], c["patterns"], vec["sigma"]
        flat = classes[L]
        assert len(flat) == L * p
        got, t = O.oracle_ac(flat, L, p, sigma, text)
        assert got == c["count_ac"] and t.patterncounter == c["distinct"]
        assert got == O.count_bruteforce(flat, L, p, text)  # the definition
        assert E.ac_scan(S.AcAutomaton.from_patterns(flat, L, p, sigma), text, blocks=2) == got
        if L >= 3:
            assert O.oracle_wu(flat, L, p, sigma, text)[0] == c["count_wu2"] == got
            assert E.wm_scan(S.WmTables.from_patterns(flat, L, p, sigma), text, blocks=3) == got
        total += got
    assert total == vec["total"]


@pytest.mark.parametrize("vec", [v for v in MIXED if min(c[0] for c in v["classes"]) >= 3],
                         ids=[v["name"] for v in MIXED if min(c[0] for c in v["classes"]) >= 3])
def test_one_pass_lane_code_reproduces_the_decomposition(vec, knob):
    """SMH_ALGO_WM sets are scanned in ONE pass: a block filter over the patterns' last min-length symbols
    proposes END columns, every survivor is verified per length class.  Same total, same positions."""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    text, patterns, lengths = cases.build_mixed(vec)
    sigma = vec["sigma"]
    classes = cases.split_classes(patterns, lengths)
    Lmin = min(classes)
    off, suf = 0, []
    for L in lengths:
        suf.append(patterns[off + int(L) - Lmin:off + int(L)])
        off += int(L)
    suffix = S.WmTables.from_patterns(np.concatenate(suf), Lmin, len(lengths), sigma)
    handles = [S.WmTables.from_patterns(classes[L], L, len(classes[L]) // L, sigma) for L in sorted(classes)]
    assert E.wm_scan_multi(suffix, handles, text, None, 2) == vec["total"]
    want_pos = np.sort(np.concatenate([O.positions_bruteforce(classes[L], L, len(classes[L]) // L, text) for L in sorted(classes)]))
    total, got = E.wm_scan_multi(suffix, handles, text, vec["total"] + 3, 3)
    assert total == vec["total"] and np.array_equal(np.sort(got).astype(np.int64), want_pos)
    if sigma == 4 and Lmin >= 8:
        # grouped pair-gram filter over the FULL patterns (two shift-or states per lane), same total and positions;
        # forced: with 200 patterns of 8 symbols the set itself takes the automaton (one column in 160 a candidate)
        assert E.build_gram_mixed(suffix, patterns, lengths, S.lib) == 1
        knob.wm("grouped=force")
        assert E.build_gram_mixed(suffix, patterns, lengths, S.lib) == 0 and suffix.info().gram_planes == 8
        for blocks in (1, 3):
            assert E.wm_scan_multi(suffix, handles, text, None, blocks) == vec["total"]
        total, got = E.wm_scan_multi(suffix, handles, text, vec["total"] + 3, 2)
        assert total == vec["total"] and np.array_equal(np.sort(got).astype(np.int64), want_pos)
    # the handle takes the one-pass form only while the suffix filter lets few columns through
    knob.wm(None)
    assert S.PatternSet(patterns, lengths, sigma, S.ALGO_WM).info().one_pass == (1 if vec["name"] in ONE_PASS | ONE_PASS_AC else 0)
    assert S.PatternSet(patterns, lengths, sigma, S.ALGO_AC).info().one_pass == (1 if vec["name"] in ONE_PASS_AC else 0)


@pytest.mark.parametrize("vec", MIXED, ids=IDS)
def test_one_pass_automaton_reproduces_the_decomposition(vec):
    """SMH_ALGO_AC sets: ONE automaton whose states carry joined (suffix-closed) output counts -- what ac/ac.c:118
    leaves out -- cut at the deepest level that fits LDS; longer patterns are verified along the goto trie.  The
    emulated lane code must give the sum of the reference's per-class counts, including the set in which short
    patterns are prefixes and suffixes of longer ones."""
    text, patterns, lengths = cases.build_mixed(vec)
    h = E.acm_compile(patterns, lengths, vec["sigma"])
    assert bool(h) == (vec["name"] in ONE_PASS_AC)
    if not h:
        assert "candidates" in S.lib.smh_last_error().decode()
        return
    for blocks in (1, 3):
        assert E.acm_scan(h, text, blocks) == vec["total"]
    assert E.acm_scan(h, text[:40], 1) == sum(O.count_bruteforce(f, L, len(f) // L, text[:40])
                                             for L, f in cases.split_classes(patterns, lengths).items() if L <= 40)
    S.lib.smh_acm_free(h)


@pytest.mark.parametrize("seed", range(8))
def test_one_pass_automaton_on_random_sets(seed):
    """random sets (nested prefixes / suffixes, duplicates, lengths 1..40, several alphabets) against the definition:
    sum over length classes of the brute-force count"""
    rng = np.random.RandomState(100 + seed)
    sigma = [2, 4, 4, 8, 20, 4, 4, 2][seed]
    n = int(rng.randint(5000, 400000))
    text = O.gen_text(n, 900 + seed, sigma)
    lengths, pats = [], []
    for j in range(int(rng.randint(2, 300))):
        L = int(rng.randint(1, 41))
        kind = rng.randint(0, 4)
        if kind == 0 and pats:  # a prefix or suffix of an earlier pattern, or the pattern again
            q = pats[rng.randint(0, len(pats))]
            L = int(rng.randint(1, len(q) + 1))
            pat = q[:L] if rng.randint(0, 2) else q[len(q) - L:]
        elif kind <= 2 and L < n:
            off = int(rng.randint(0, n - L))
            pat = text[off:off + L]
        else:
            pat = rng.randint(0, sigma, size=L).astype(np.uint8)
        pats.append(np.array(pat, dtype=np.uint8))
        lengths.append(len(pat))
    patterns = np.concatenate(pats)
    want = sum(O.count_bruteforce(f, L, len(f) // L, text) for L, f in cases.split_classes(patterns, np.array(lengths)).items())
    h = E.acm_compile(patterns, lengths, sigma)
    if not h:
        pytest.skip(S.lib.smh_last_error().decode())
    assert E.acm_scan(h, text, 2) == want
    S.lib.smh_acm_free(h)


@pytest.mark.parametrize("algo", [S.ALGO_AC, S.ALGO_WM])
def test_set_handle_groups_by_length(algo):
    vec = MIXED[0]
    _, patterns, lengths = cases.build_mixed(vec)
    ps = S.PatternSet(patterns, lengths, vec["sigma"], algo)
    info = ps.info()
    assert ps.classes() == [tuple(c) for c in vec["classes"]]
    assert (info.classes, info.patterns, info.min_length, info.max_length, info.algorithm) == \
           (len(vec["classes"]), len(lengths), vec["classes"][0][0], vec["classes"][-1][0], algo)
    ps.close()


def test_set_compile_rejects_bad_input():
    with pytest.raises(S.SmhError, match="length 0"):
        S.PatternSet(np.array([1, 2], dtype=np.uint8), [2, 0], 4)
    with pytest.raises(S.SmhError, match="alphabet"):
        S.PatternSet(np.array([1, 2, 9], dtype=np.uint8), [1, 2], 4)
    with pytest.raises(S.SmhError, match="lengths sum"):
        S.PatternSet(np.array([1, 2, 3], dtype=np.uint8), [1, 1], 4)
    with pytest.raises(S.SmhError, match="bad arguments"):
        S.PatternSet(np.array([1, 2, 3], dtype=np.uint8), [1, 2], 4, algorithm=7)
    if S.device_count() == 0:  # no GPU: the scan fails loudly, nothing is computed on the host
        ps = S.PatternSet(np.array([0, 1, 2, 3, 1, 2], dtype=np.uint8), [4, 2], 4)
        with pytest.raises(S.SmhError):
            ps.count_host(np.zeros(100, dtype=np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("vec", MIXED, ids=IDS)
def test_gpu_set_counts_match_reference_decomposition(vec):
    import torch
    text, patterns, lengths = cases.build_mixed(vec)
    n, sigma = len(text), vec["sigma"]
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(text).to(dev)
    classes = cases.split_classes(patterns, lengths)
    want_pos = np.sort(np.concatenate([O.positions_bruteforce(classes[L], L, len(classes[L]) // L, text)
                                       for L in sorted(classes)]))
    assert len(want_pos) == vec["total"]
    for algo in (S.ALGO_AC, S.ALGO_WM):
        ps = S.PatternSet(patterns, lengths, sigma, algo)
        assert ps.count_host(text)[0] == vec["total"]
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        ps.scan_device(d_text.data_ptr(), n, cnt.data_ptr(), stream)
        ps.scan_device(d_text.data_ptr(), n, cnt.data_ptr(), stream)  # accumulates
        cap = vec["total"] + 3
        pos = torch.zeros(cap, dtype=torch.int64, device=dev)
        ps.positions_device(d_text.data_ptr(), n, pos.data_ptr(), cap, cnt.data_ptr() + 8, stream)
        torch.cuda.synchronize()
        assert cnt.tolist() == [2 * vec["total"], vec["total"]]
        assert np.array_equal(np.sort(pos[:vec["total"]].cpu().numpy()), want_pos)
        ps.close()


@pytest.mark.gpu
def test_gpu_set_at_baseline_shape():
    """BASELINE configs[1] read as ONE set: 1000 DNA patterns with lengths drawn from 8..32, 64 MiB of text;
    the set count equals the sum of fixed-length counts of its classes (each checked against the oracle on a
    prefix) and AC == WM."""
    sigma, n = 4, 64 << 20
    text = S.corpus_text(n, 42, sigma)
    rng = np.random.RandomState(5)
    lengths = rng.randint(8, 33, size=1000).astype(np.uint32)
    pats = []
    for j, L in enumerate(lengths):
        if j % 2 == 0:
            off = int(rng.randint(0, n - L))
            pats.append(text[off:off + L])
        else:
            pats.append(rng.randint(0, sigma, size=L).astype(np.uint8))
    patterns = np.concatenate(pats)
    ac = S.PatternSet(patterns, lengths, sigma, S.ALGO_AC)
    wm = S.PatternSet(patterns, lengths, sigma, S.ALGO_WM)
    got = ac.count_host(text)[0]
    assert got == wm.count_host(text)[0] and got >= 500
    sample = text[:4 << 20]
    want = sum(O.oracle_ac(flat, L, len(flat) // L, sigma, sample)[0]
               for L, flat in cases.split_classes(patterns, lengths).items())
    assert ac.count_host(sample)[0] == want == wm.count_host(sample)[0]


def _random_dna_set(seed):
    """patterns of lo..hi symbols in two plane groups (shorter than 14 / 14 and more), suffixes of one another,
    duplicates, occurrences planted across segment and wave-chunk boundaries and at both ends of the text"""
    rng = np.random.RandomState(500 + seed)
    n = int(rng.randint(3 * 4096 + 100, 6 * 4096))
    text = rng.randint(0, 4, size=n).astype(np.uint8)
    lo = [8, 8, 9, 10, 13, 14, 8, 11, 8, 16][seed]
    hi = [32, 13, 40, 10, 33, 40, 9, 12, 20, 17][seed]
    pats, lengths = [], []
    for j in range(int(rng.randint(2, 400))):
        L = int(rng.randint(lo, hi + 1))
        if j % 5 == 4 and pats:  # a suffix of an earlier pattern (same end, shorter), or the pattern again
            q = pats[rng.randint(0, len(pats))]
            L = int(rng.randint(lo, len(q) + 1)) if len(q) >= lo else len(q)
            pat = q[len(q) - L:]
        else:
            pat = rng.randint(0, 4, size=L).astype(np.uint8)
        pats.append(np.array(pat, dtype=np.uint8))
        lengths.append(len(pat))
    if len(set(lengths)) < 2:
        pats.append(rng.randint(0, 4, size=lo + 1).astype(np.uint8)); lengths.append(lo + 1)
    for i, off in enumerate([0, 300, 640 - 5, 4096 - 7, 4096 - 1, 8191, 8192, 8192 + 100, n - 1]):
        q = pats[(3 * i) % len(pats)]
        off = min(max(off - (len(q) if off == n - 1 else 0) + (1 if off == n - 1 else 0), 0), n - len(q))
        text[off:off + len(q)] = q
    patterns, lengths = np.concatenate(pats), np.array(lengths, dtype=np.uint32)
    classes = cases.split_classes(patterns, lengths)
    want_pos = np.sort(np.concatenate([O.positions_bruteforce(classes[L], L, len(classes[L]) // L, text) for L in sorted(classes)]))
    assert len(want_pos) >= 5
    return text, patterns, lengths, classes, want_pos


@pytest.mark.parametrize("tune", ["grouped=force", "grouped=force,split14"], ids=["split_chosen", "split_at_14"])
@pytest.mark.parametrize("seed", range(10))
def test_grouped_pair_gram_filter_on_random_dna_sets(seed, tune, knob):
    """the one-pass form of SMH_ALGO_WM sets on the 4-letter alphabet (emulated lane code) against the definition
    (sum over length classes); the two groups split where the builder estimates the fewest candidates (round 5) and at the
    fixed length of round 4"""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    text, patterns, lengths, classes, want_pos = _random_dna_set(seed)
    want = len(want_pos)
    Lmin = min(classes)
    off, suf = 0, []
    for L in lengths:
        suf.append(patterns[off + int(L) - Lmin:off + int(L)])
        off += int(L)
    suffix = S.WmTables.from_patterns(np.concatenate(suf), Lmin, len(lengths), 4)
    handles = [S.WmTables.from_patterns(classes[L], L, len(classes[L]) // L, 4) for L in sorted(classes)]
    knob.wm(tune)  # whatever the candidate rate: the count must not depend on it
    assert E.build_gram_mixed(suffix, patterns, lengths, S.lib) == 0
    assert E.wm_scan_multi(suffix, handles, text, None, 2) == want
    total, got = E.wm_scan_multi(suffix, handles, text, want + 3, 3)
    assert total == want and np.array_equal(np.sort(got).astype(np.int64), want_pos)
    knob.wm(None)
    assert S.PatternSet(patterns, lengths, 4, S.ALGO_WM).info().one_pass == 1  # grouped filter or automaton


@pytest.mark.gpu
@pytest.mark.parametrize("force", [True, False], ids=["grouped_forced", "as_chosen"])
@pytest.mark.parametrize("seed", range(10))
def test_gpu_grouped_pair_gram_filter_on_random_dna_sets(seed, force, knob):
    """the same sets through smh_pset_* on the device: count and positions of SMH_ALGO_WM sets, with the grouped
    pair-gram filter forced and with whatever one-pass form the set chose"""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    import torch
    text, patterns, lengths, classes, want_pos = _random_dna_set(seed)
    want, n = len(want_pos), len(text)
    if force:
        knob.wm("grouped=force")
    ps = S.PatternSet(patterns, lengths, 4, S.ALGO_WM)
    assert ps.info().one_pass == 1
    assert ps.count_host(text)[0] == want
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(text).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    pos = torch.zeros(want + 3, dtype=torch.int64, device=dev)
    ps.positions_device(d_text.data_ptr(), n, pos.data_ptr(), want + 3, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cnt.item()) == want and np.array_equal(np.sort(pos[:want].cpu().numpy()), want_pos)
    ps.close()


def _nested_suffix_set(seed=5):
    """patterns that share their LAST EIGHT symbols -- one slot of the verify stage's suffix index, a chain of records -- some
    of them suffixes of others (several patterns end at one column), lengths on both sides of 16 and 32 symbols, a duplicate"""
    rng = np.random.RandomState(seed)
    n = 3 * 4096 + 777
    text = rng.randint(0, 4, size=n).astype(np.uint8)
    long = rng.randint(0, 4, size=45).astype(np.uint8)
    pats = [long[45 - L:] for L in (8, 9, 12, 16, 17, 31, 32, 33, 40, 45)]       # nested: each a suffix of the next
    other = rng.randint(0, 4, size=20).astype(np.uint8)
    other[-8:] = long[-8:]                                                       # same slot, not nested
    pats += [other, other[3:], other.copy(), rng.randint(0, 4, size=14).astype(np.uint8), rng.randint(0, 4, size=26).astype(np.uint8)]
    for off in (0, 10, 500, 4096 - 45, 4096 - 20, 4096 + 30, 2 * 4096 - 3, n - 45):
        text[off:off + 45] = long
    for off in (100, 4096 - 9, 9000, n - 20):
        text[off:off + 20] = other
    text[3:3 + 12] = long[45 - 12:]                                              # a short one in the text's first bytes
    text[7000:7000 + 14] = pats[-2]
    text[7100:7100 + 26] = pats[-1]
    patterns, lengths = np.concatenate(pats), np.array([len(q) for q in pats], dtype=np.uint32)
    classes = cases.split_classes(patterns, lengths)
    want_pos = np.sort(np.concatenate([O.positions_bruteforce(classes[L], L, len(classes[L]) // L, text) for L in sorted(classes)]))
    return text, patterns, lengths, classes, want_pos


@pytest.mark.parametrize("tune", ["grouped=force", "grouped=force,sfx=0"], ids=["suffix_index", "class_by_class"])
def test_suffix_index_chains_nested_and_long_patterns(tune, knob):
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    text, patterns, lengths, classes, want_pos = _nested_suffix_set()
    want = len(want_pos)
    assert want > 50 and len(np.unique(want_pos)) < want  # several patterns end at one column
    Lmin = min(classes)
    off, suf = 0, []
    for L in lengths:
        suf.append(patterns[off + int(L) - Lmin:off + int(L)])
        off += int(L)
    suffix = S.WmTables.from_patterns(np.concatenate(suf), Lmin, len(lengths), 4)
    handles = [S.WmTables.from_patterns(classes[L], L, len(classes[L]) // L, 4) for L in sorted(classes)]
    knob.wm(tune)
    assert E.build_gram_mixed(suffix, patterns, lengths, S.lib) == 0
    assert E.wm_scan_multi(suffix, handles, text, None, 2) == want
    total, got = E.wm_scan_multi(suffix, handles, text, want + 3, 3)
    assert total == want and np.array_equal(np.sort(got).astype(np.int64), want_pos)


@pytest.mark.gpu
@pytest.mark.parametrize("tune", ["grouped=force", "grouped=force,sfx=0"], ids=["suffix_index", "class_by_class"])
def test_gpu_suffix_index_chains_nested_and_long_patterns(tune, knob):
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    import torch
    text, patterns, lengths, classes, want_pos = _nested_suffix_set()
    want, n = len(want_pos), len(text)
    knob.wm(tune)
    ps = S.PatternSet(patterns, lengths, 4, S.ALGO_WM)
    assert ps.info().one_pass == 1
    assert ps.count_host(text)[0] == want
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(text).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    pos = torch.zeros(want + 3, dtype=torch.int64, device=dev)
    ps.positions_device(d_text.data_ptr(), n, pos.data_ptr(), want + 3, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cnt.item()) == want and np.array_equal(np.sort(pos[:want].cpu().numpy()), want_pos)
    ps.close()


def _wide_dna_set(p=2400, lo=8, hi=40, n=1 << 20, seed=77):
    """thousands of DNA patterns over a wide range of lengths: too many short ones for the grouped filter, too many
    nodes for one automaton in LDS"""
    rng = np.random.RandomState(seed)
    text = rng.randint(0, 4, size=n).astype(np.uint8)
    lengths = rng.randint(lo, hi + 1, size=p).astype(np.uint32)
    pats = []
    for j, L in enumerate(lengths):
        if j % 3 == 0:
            off = int(rng.randint(0, n - L))
            pats.append(text[off:off + L])
        else:
            pats.append(rng.randint(0, 4, size=L).astype(np.uint8))
    return text, np.concatenate(pats), lengths


def test_split_form_is_chosen_for_large_wide_sets():
    text, patterns, lengths = _wide_dna_set()
    wm = S.PatternSet(patterns, lengths, 4, S.ALGO_WM)
    info = wm.info()
    assert info.one_pass == 0 and info.passes == 2 and info.classes == 33   # long patterns: filter pass; short: automaton pass
    ac = S.PatternSet(patterns, lengths, 4, S.ALGO_AC)
    assert ac.info().one_pass == 0 and ac.info().passes == 33
    small = S.PatternSet(patterns[:int(lengths[:50].sum())], lengths[:50], 4, S.ALGO_WM)
    assert small.info().one_pass == 1 and small.info().passes == 1


@pytest.mark.gpu
def test_gpu_split_form_counts_and_positions():
    import torch
    text, patterns, lengths = _wide_dna_set()
    n = len(text)
    classes = cases.split_classes(patterns, lengths)
    want_pos = np.sort(np.concatenate([O.positions_bruteforce(classes[L], L, len(classes[L]) // L, text) for L in sorted(classes)]))
    want = len(want_pos)
    assert want > 1000
    ps = S.PatternSet(patterns, lengths, 4, S.ALGO_WM)
    assert ps.info().passes == 2
    assert ps.count_host(text)[0] == want
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(text).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    pos = torch.zeros(want + 3, dtype=torch.int64, device=dev)
    ps.positions_device(d_text.data_ptr(), n, pos.data_ptr(), want + 3, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cnt.item()) == want and np.array_equal(np.sort(pos[:want].cpu().numpy()), want_pos)
    ps.close()
