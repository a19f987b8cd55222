"""The window-hash engine (round 5; csrc/hash_engine.h, hash_host.c, hash_lane.h, hash_kernels.hip): a Bloom filter of the rolling
hash of the WHOLE m-byte window in LDS -- a pass rate that does not depend on the text -- and the patterns themselves in a cuckoo
hash in device memory.  Held by Wu-Manber handles over byte-like alphabets whose set the key engine does not take.

CPU: the lane code (compiled for the CPU, tests/emu) against the brute-force definition and the oracle; GPU (-m gpu): the kernel
forced inside the handles against the same and the reference's golden vectors."""
import json
import os

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O
import smatcher_hip as S

HERE = os.path.dirname(os.path.abspath(__file__))
# (alphabet, m, patterns): every slot size / window-dword count / out-stream alignment ((4 - m) & 3), both filter sizes
SETS = [(256, 9, 300), (256, 10, 3000), (256, 11, 500), (256, 12, 2000), (256, 16, 800), (256, 17, 400), (256, 20, 5000), (256, 31, 200), (256, 32, 300),
        (128, 10, 500), (20, 13, 600), (20, 16, 1000), (20, 29, 100)]


def _case(sigma, m, p, n, seed=42):
    text = O.gen_text(n, seed, sigma)
    return text, O.gen_patterns_mixed(m, p, 7, sigma, seed, n, 2)


@pytest.mark.parametrize("sigma,m,p", SETS)
def test_lane_code_counts_what_the_definition_counts(sigma, m, p):
    n = 3 * 4096 + 2345
    text, pat = _case(sigma, m, p, n)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    info = wm.info()
    assert info.hash_slots > 0 and info.key_slots == 0 and info.adaptive == 1
    want = O.count_bruteforce(pat, m, p, text)
    assert want > 0
    got, passed = E.hash_scan(wm, text)
    assert got == want
    assert want <= passed <= want + max(40, int(0.08 * n))  # the filter: every match, and a text-independent few per cent besides
    assert E.hash_scan(wm, text, blocks=1)[0] == want
    wm.close()


@pytest.mark.parametrize("n", [0, 5, 11, 12, 13, 63, 64, 65, 4095, 4096, 4097, 4159, 4160, 4161, 8191, 8192, 8256, 8257, 12288 + 40])
def test_text_length_edges(n):
    sigma, m, p = 256, 12, 100
    text, pat = _case(sigma, m, p, max(n, 64))
    text = text[:n]
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert E.hash_scan(wm, text)[0] == O.count_bruteforce(pat, m, p, text)
    wm.close()


def test_matches_across_every_lane_and_chunk_boundary():
    for sigma, m in ((256, 12), (256, 20), (256, 32), (20, 13)):
        unit = O.gen_text(m, 5, sigma)
        text = np.tile(unit, (5 * 4096) // m + 2)[:5 * 4096 + 77]
        rot = np.concatenate([np.roll(unit, -r) for r in range(m)])
        wm = S.WmTables.from_patterns(rot, m, m, sigma)
        assert wm.info().hash_slots > 0
        got, passed = E.hash_scan(wm, text)
        assert got == len(text) - m + 1 == O.count_bruteforce(rot, m, m, text) and passed == got
        wm.close()


def test_filter_rate_does_not_depend_on_the_text():
    """100 000 patterns of 12 bytes sampled from natural-language-like text: their 3-byte grams are the text's common grams, so a gram
    filter passes most columns of that text -- the window-hash filter passes the matches and the table's few per cent"""
    n, m, p, sigma = 1 << 18, 12, 100000, 256
    text = S.corpus_text(n, 42, sigma, 0, S.CORPUS_SKEWED)
    pat = S.corpus_patterns(m, p, 12, sigma, 42, n, 2, S.CORPUS_SKEWED)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert wm.info().hash_slots >= wm.info().distinct > 80000 and wm.adapt().est_ms_per_gib[S.ENGINE_HASH] > 0.3  # two-slot buckets, 82 % full
    want = O.count_bruteforce(pat, m, p, text[:1 << 16])
    got, passed = E.hash_scan(wm, text[:1 << 16])
    assert got == want
    assert passed - want < 0.07 * (1 << 16), (passed, want)
    uni = O.gen_text(1 << 16, 43, sigma)
    got_u, passed_u = E.hash_scan(wm, uni)
    assert got_u == O.count_bruteforce(pat, m, p, uni) and passed_u < 0.07 * (1 << 16)
    wm.close()


def test_positions_are_the_end_columns():
    sigma, m, p, n = 256, 12, 300, 3 * 4096 + 100
    text, pat = _case(sigma, m, p, n)
    want = np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.uint64)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    total, got = E.hash_positions(wm, text, len(want) + 8)
    assert total == len(want) and np.array_equal(np.sort(got), np.sort(want))
    wm.close()


def test_which_handles_keep_it():
    for (sigma, m, p), want in (((256, 8, 3000), False),     # the key engine takes it
                                ((256, 8, 100000), True),    # more keys than LDS holds
                                ((256, 12, 1000), True),     # 96-bit keys
                                ((20, 16, 1000), True),
                                ((4, 16, 1000), False)):     # the 4-letter alphabet has the key engine / the flat parts
        wm = S.WmTables.from_patterns(O.gen_patterns(m, p, 7, sigma), m, p, sigma)
        assert (wm.info().hash_slots > 0) == want, (sigma, m, p)
        if want:
            wm.set_scan_engine(S.ENGINE_HASH)
            assert wm.info().scan_engine == S.ENGINE_HASH
        else:
            with pytest.raises(S.SmhError, match="window-hash"):
                wm.set_scan_engine(S.ENGINE_HASH)
        wm.close()


def test_golden_vectors_of_the_reference_through_the_lane_code():
    import cases
    vectors = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
    taken = 0
    for v in vectors:
        if v["sigma"] not in (8, 20, 128, 256) or v["m"] < 4 or v["n"] > 130000:
            continue
        text, pat = cases.build(v)
        wm = S.WmTables.from_patterns(pat, v["m"], v["p"], v["sigma"])
        if wm.info().hash_slots:
            assert E.hash_scan(wm, text, blocks=1)[0] == v["count_wu"], v["name"]
            taken += 1
        wm.close()
    assert taken >= 8, taken


@pytest.mark.parametrize("sigma,m,p", [(256, 12, 2000), (256, 20, 5000), (256, 9, 300), (20, 16, 1000), (256, 32, 300)])
def test_three_filter_bits_pass_fewer_windows_and_count_the_same(sigma, m, p, knob):
    """Round 6: a third bit per window in the same filter word (hash_engine.h bloom_k = 3; testing twin: the count of bits is a
    development knob).  Same count, every match still passes the filter, fewer non-matching windows do."""
    T = knob.T
    n = 3 * 4096 + 2345
    text, pat = _case(sigma, m, p, n)
    want = O.count_bruteforce(pat, m, p, text)
    passed = {}
    for bits in (2, 3):
        knob.set(T.TUNE_HASH, "bits=%d" % bits)
        wm = T.WmTables.from_patterns(pat, m, p, sigma)
        got, passed[bits] = E.hash_scan(wm, text)
        assert got == want and passed[bits] >= want, bits
        assert E.hash_scan(wm, text, blocks=1)[0] == want
        total, pos = E.hash_positions(wm, text, want + 8) if hasattr(E, "hash_positions") else (want, None)
        assert total == want
        wm.close()
    assert passed[3] <= passed[2]
