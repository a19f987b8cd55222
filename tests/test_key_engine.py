"""The key engine (round 5; csrc/key_hash.h, key_host.c, key_lane.h, key_kernels.hip): the patterns of ONE length as a two-table
cuckoo hash of their keys in LDS, one exact membership test per text column.

CPU: the host builder and the kernels' lane code (compiled for the CPU, tests/emu) against the oracle's search_ac restatement and
a brute-force count on every alphabet / length class the engine takes, tiling edges included.  GPU (-m gpu): the real kernels
through the C ABI against the same, the reference's golden vectors, and the engine forced inside smh_ac / smh_wm handles."""
import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O
import smatcher_hip as S

# (alphabet, m, patterns): 32-bit keys, 64-bit keys, every symbol width, the halo classes (m - 1 <= 16 / <= 32), key widths that fill the slot
SETS = [(4, 17, 1200), (4, 20, 300), (4, 21, 5000), (20, 7, 400), (20, 8, 10000), (256, 5, 3000), (128, 5, 100), (128, 6, 200), (8, 14, 700),  # quotient keys (33..42 bits)
        (4, 3, 10), (4, 8, 100), (4, 16, 1000), (4, 17, 300), (4, 32, 500), (2, 16, 40), (8, 10, 200), (8, 21, 100), (20, 6, 300),
        (20, 8, 500), (20, 12, 200), (128, 4, 300), (128, 9, 100), (256, 3, 50), (256, 4, 1000), (256, 5, 400), (256, 8, 2000), (16, 16, 100)]


def _text_and_patterns(sigma, m, p, n, seed=42):
    text = O.gen_text(n, seed, sigma)
    pat = O.gen_patterns_mixed(m, p, 7, sigma, seed, n, 2)  # every second pattern is a substring of the text
    return text, pat


@pytest.mark.parametrize("sigma,m,p", SETS)
def test_builder_holds_exactly_the_set(sigma, m, p):
    text, pat = _text_and_patterns(sigma, m, p, 1 << 14)
    k = S.KeyTable(pat, m, p, sigma)
    info = k.info()
    pats = np.asarray(pat, dtype=np.uint8).reshape(p, m)
    assert info.keys == len({bytes(r) for r in pats})
    assert info.key_bits == m * max(2, int(np.ceil(np.log2(sigma)))) and info.slot_bytes == (8 if info.key_bits > 42 else 4)  # 33..42 bits: quotient keys
    assert info.keys <= 0.485 * 2 * info.slots and info.lds_bytes <= 156 * 1024
    k.close()


@pytest.mark.parametrize("sigma,m,p", SETS)
def test_lane_code_counts_what_search_ac_counts(sigma, m, p):
    # 3 wave-chunks and a ragged tail: chunk 0 and the last chunk take the bounds-checked path, the middle one the register path
    n = 3 * 4096 + 1234
    text, pat = _text_and_patterns(sigma, m, p, n)
    want = O.count_bruteforce(pat, m, p, text)
    assert want > 0
    k = S.KeyTable(pat, m, p, sigma)
    assert E.keys_scan(k, text) == want
    assert E.keys_scan(k, text, blocks=1) == want
    k.close()


@pytest.mark.parametrize("n", [0, 1, 7, 8, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 12289])
def test_text_length_edges(n):
    sigma, m, p = 4, 8, 64
    text, pat = _text_and_patterns(sigma, m, p, max(n, 64))
    text = text[:n]
    k = S.KeyTable(pat, m, p, sigma)
    assert E.keys_scan(k, text) == O.count_bruteforce(pat, m, p, text)
    k.close()


def test_matches_across_every_lane_and_chunk_boundary():
    """a text that is ONE pattern repeated: every column from m - 1 on ends a match, across segment, wave-chunk and workgroup edges"""
    for sigma, m in ((4, 16), (4, 32), (256, 8), (20, 12), (20, 8), (4, 19), (256, 5)):
        unit = O.gen_text(m, 5, sigma)
        text = np.tile(unit, (5 * 4096) // m + 2)[:5 * 4096 + 77]
        rot = np.concatenate([np.roll(unit, -r) for r in range(m)])  # all rotations of the unit
        k = S.KeyTable(rot, m, m, sigma)
        assert E.keys_scan(k, text) == len(text) - m + 1 == O.count_bruteforce(rot, m, m, text)
        k.close()


def test_agrees_with_the_restated_search_ac_and_search_wu():
    for sigma, m, p in ((4, 8, 100), (4, 16, 300), (256, 8, 200)):
        text, pat = _text_and_patterns(sigma, m, p, 1 << 15)
        want, _ = O.oracle_ac(pat, m, p, sigma, text)
        k = S.KeyTable(pat, m, p, sigma)
        assert E.keys_scan(k, text) == want
        if sigma in (4, 256):
            assert O.oracle_wu(pat, m, p, sigma, text)[0] == want
        k.close()


def test_positions_are_the_end_columns():
    sigma, m, p, n = 4, 12, 200, 3 * 4096 + 100
    text, pat = _text_and_patterns(sigma, m, p, n)
    want = np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.uint64)
    k = S.KeyTable(pat, m, p, sigma)
    total, got = E.keys_positions(k, text, len(want) + 8)
    assert total == len(want) and np.array_equal(np.sort(got), np.sort(want))
    k.close()


def test_a_key_that_fills_its_slot_never_matches_a_free_slot():
    """m * bits == 32 / 64: every slot value is a possible key, so free slots hold keys that hash elsewhere -- a text made of exactly
    those filler values must count nothing"""
    for sigma, m in ((4, 16), (256, 4), (4, 32), (256, 8), (256, 5), (4, 20), (20, 8)):
        pat = O.gen_patterns(m, 3, 11, sigma)
        k = S.KeyTable(pat, m, 3, sigma)
        bits = {4: 2, 20: 5, 256: 8}[sigma]
        fillers = []
        # the builder's fillers are small numbers; quotient keys: any high bits in front of them
        for v in list(range(64)) + ([(y << 32) | x for y in (1, 2, 3, (1 << (bits * m - 32)) - 1) for x in range(16)] if 32 < bits * m <= 42 else []):
            sym = [(v >> (bits * (m - 1 - i))) & ((1 << bits) - 1) if bits * (m - 1 - i) < 64 else 0 for i in range(m)]
            if max(sym) < sigma:
                fillers.append(np.asarray(sym, dtype=np.uint8))
        text = np.concatenate(fillers * 40)
        assert E.keys_scan(k, text) == O.count_bruteforce(pat, m, 3, text)
        k.close()


def test_sets_the_engine_does_not_take():
    with pytest.raises(S.SmhError, match="64"):
        S.KeyTable(O.gen_patterns(33, 10, 7, 4), 33, 10, 4)
    with pytest.raises(S.SmhError, match="64"):
        S.KeyTable(O.gen_patterns(9, 10, 7, 256), 9, 10, 256)
    with pytest.raises(S.SmhError, match="LDS"):
        S.KeyTable(O.gen_patterns(8, 30000, 7, 256), 8, 30000, 256)
    S.KeyTable(O.gen_patterns(8, 16000, 7, 20), 8, 16000, 20).close()   # 40-bit keys as quotient keys in 4-byte slots: twice the 64-bit capacity
    S.KeyTable(O.gen_patterns(32, 8000, 7, 4), 32, 8000, 4).close()   # BASELINE configs[3]'s longest set fits
    S.KeyTable(O.gen_patterns(16, 16000, 7, 4), 16, 16000, 4).close()


def test_golden_vectors_of_the_reference_through_the_lane_code():
    """the counts the reference's own search_ac produced (tests/golden/ref_vectors.json), every vector the engine takes"""
    import json
    import os
    import cases
    vectors = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.json")))
    taken = 0
    for v in vectors:
        if v["m"] * max(2, int(np.ceil(np.log2(v["sigma"])))) > 64 or v["n"] > 130000:
            continue
        text, pat = cases.build(v)
        k = S.KeyTable(pat, v["m"], v["p"], v["sigma"])
        assert E.keys_scan(k, text, blocks=1) == v["count_ac"], v["name"]
        k.close()
        taken += 1
    assert taken > 90


def test_which_handles_keep_a_key_table():
    """held beside every plan or path whose rate depends on the text or that is verify-bound -- not beside an exact one-launch plan,
    not beside a set ONE plain stride-1 image serves (0.25 ms/GiB: faster than two LDS reads per column)"""
    # (alphabet, m, patterns) -> key table expected through the (AC, WM) entry points
    cases = [((4, 8, 1000), (False, False)),      # dense plan / pair table: exact, one launch
             ((4, 16, 1000), (False, False)),     # one plain stride-1 image holds the set
             ((4, 32, 1000), (True, True)),       # two parts
             ((4, 16, 8000), (True, True)),
             ((4, 32, 8000), (True, True)),
             ((20, 8, 1000), (True, True)),
             ((256, 8, 3000), (True, True)),      # verify-bound automaton plan / byte-gram filter
             ((256, 12, 3000), (False, False)),   # 96 bits: not a set the engine takes
             ((256, 8, 100000), (False, False))]  # more keys than LDS holds
    for (sigma, m, p), (want_ac, want_wm) in cases:
        pat = O.gen_patterns(m, p, 7, sigma)
        if p <= 8000:
            ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
            assert (ac.info().key_slots > 0) == want_ac, ("ac", sigma, m, p, ac.info().key_slots)
            if want_ac:
                assert ac.info().adaptive == 1 and ac.adapt().est_ms_per_gib[S.ENGINE_KEYS] > 0.2
                ac.set_scan_engine(S.ENGINE_KEYS)
                assert ac.info().scan_engine == S.ENGINE_KEYS and ac.info().adaptive == 0
                ac.set_scan_engine(-1)
            else:
                with pytest.raises(S.SmhError, match="key table"):
                    ac.set_scan_engine(S.ENGINE_KEYS)
            ac.close()
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        assert (wm.info().key_slots > 0) == want_wm, ("wm", sigma, m, p, wm.info().key_slots)
        if want_wm:
            assert wm.info().adaptive == 1 and wm.adapt().est_ms_per_gib[S.ENGINE_KEYS] > 0.2
            wm.set_scan_engine(S.ENGINE_KEYS)
            assert wm.info().scan_engine == S.ENGINE_KEYS
        else:
            with pytest.raises(S.SmhError, match="key table"):
                wm.set_scan_engine(S.ENGINE_KEYS)
        wm.close()


# ---- round 6: the bucket image (csrc/key_hash.h "The bucket image"): one 8-byte LDS read per column ----
# (alphabet, m, patterns): window within the 32-bit image (with and without spare low bits), DNA windows of 17..21 symbols and protein
# windows of 7 / 8 (the older symbols come out of the delay line), sets big enough to fill the overflow table
BUCKET_SETS = [(4, 16, 1000), (4, 16, 8000), (4, 16, 10000), (4, 12, 3000), (4, 10, 500), (4, 17, 1200), (4, 19, 4000), (4, 20, 5000),
               (20, 6, 300), (20, 7, 400), (20, 8, 10000), (8, 10, 200), (16, 8, 100), (128, 4, 300), (256, 3, 5000), (256, 4, 1000), (2, 16, 40)]


def _contains(lib, k, key):
    lib.smh_keys_contains.restype = S.C.c_int
    lib.smh_keys_contains.argtypes = [S.C.c_void_p, S.C.c_uint64]
    return int(lib.smh_keys_contains(k.h, key))


@pytest.mark.parametrize("sigma,m,p", BUCKET_SETS)
def test_bucket_image_holds_exactly_the_set_and_counts_what_the_definition_counts(sigma, m, p, knob):
    T = knob.T
    n = 3 * 4096 + 1234
    text, pat = _text_and_patterns(sigma, m, p, n)
    want = O.count_bruteforce(pat, m, p, text)
    bits = max(2, int(np.ceil(np.log2(sigma))))
    keys = {int("".join(format(int(c), "0%db" % bits) for c in row), 2) for row in np.asarray(pat, dtype=np.uint8).reshape(p, m)}
    rng = np.random.RandomState(m * 1000 + p)
    counts = {}
    for layout in (1, 0):
        knob.set(T.TUNE_KEY, "layout=%d" % layout)
        k = T.KeyTable(pat, m, p, sigma)
        info = k.info()
        assert info.layout == layout and info.keys == len(keys) and info.lds_bytes <= 156 * 1024
        assert all(_contains(T.lib, k, key) for key in list(keys)[:2000])
        # near misses: a key with one symbol changed is in the set only if it is another key
        for key in list(keys)[:300]:
            for _ in range(4):
                pos, sym = int(rng.randint(m)), int(rng.randint(sigma))
                other = (key & ~(((1 << bits) - 1) << (bits * pos))) | (sym << (bits * pos))
                assert _contains(T.lib, k, other) == (other in keys)
        counts[layout] = (E.keys_scan(k, text), E.keys_scan(k, text, blocks=1))
        if layout == 1 and p >= 8000:
            assert info.overflow_keys > 0  # buckets of three keys and more exist: the sentinel path ran
        total, got = E.keys_positions(k, text, want + 8)
        assert total == want and np.array_equal(np.sort(got), np.sort(np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.uint64)))
        k.close()
    assert counts[0] == counts[1] == (want, want) and want > 0


def test_bucket_image_is_the_default_where_it_is_the_faster_one():
    """sets of up to ~4000 keys whose window the image takes (alphabet 4: 10..20 symbols; 20 letters: up to 8): few crowded buckets,
    0.29-0.37 ms/GiB against the cuckoo image's 0.41-0.46; from ~4500 keys up the crowded buckets' share (8000 keys: 1.4 %) costs
    more than the second LDS read saves (0.53 against 0.41: profiles/r06_final/notes/ab_key_bucket_image.log) -- the builder measures the
    share and keeps the cuckoo image there"""
    for (sigma, m, p), layout in (((4, 16, 1500), 1), ((4, 16, 3500), 1), ((20, 8, 3000), 1), ((4, 20, 3000), 1), ((4, 16, 8000), 0), ((20, 8, 10000), 0),
                                  ((4, 32, 3000), 0), ((256, 8, 2000), 0), ((4, 8, 100), 0), ((20, 12, 200), 0)):
        k = S.KeyTable(O.gen_patterns(m, p, 7, sigma), m, p, sigma)
        info = k.info()
        assert info.layout == layout, (sigma, m, p)
        assert (0.28 < info.est_ms_per_gib < 0.40) if layout == 1 else info.est_ms_per_gib >= 0.40
        k.close()


def test_bucket_image_with_the_all_zero_window_and_crowded_buckets(knob):
    """poly-A: the window whose image H is 0 -- the sentinel's value when the image has no spare bits -- as a pattern, alone in its
    bucket, beside one other key, and in a crowded bucket; and a set of near-identical patterns (one symbol apart: the same last 15
    symbols land in one bucket) that pushes keys through the sentinel into the overflow table"""
    T = knob.T
    knob.set(T.TUNE_KEY, "layout=1")
    sigma, m = 4, 16
    zero = np.zeros(m, dtype=np.uint8)
    rng = np.random.RandomState(5)
    base = rng.randint(0, sigma, size=(200, m)).astype(np.uint8)
    family = []
    for row in base:  # all four choices of the OLDEST symbol: same bucket bits from the 15 newer ones as far as the multiplier lets them
        for s in range(sigma):
            r = row.copy(); r[0] = s; family.append(r)
    for extra in ([], [base[0]], family):
        pats = np.stack([zero] + list(extra)) if len(extra) else zero[None, :]
        p = len(pats)
        text = np.concatenate([np.zeros(500, dtype=np.uint8), np.concatenate(list(pats) * 3), rng.randint(0, sigma, size=9000).astype(np.uint8), np.zeros(300, dtype=np.uint8)])
        k = T.KeyTable(pats.reshape(-1), m, p, sigma)
        assert k.info().layout == 1
        want = O.count_bruteforce(pats.reshape(-1), m, p, text)
        assert E.keys_scan(k, text) == want and want >= 500 - m + 1
        k.close()


def test_golden_vectors_through_both_images(knob):
    """the reference's own counts (tests/golden/ref_vectors.json) through the bucket image wherever it applies, and through the
    cuckoo image forced"""
    import json
    import os
    import cases
    T = knob.T
    vectors = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.json")))
    ran = {0: 0, 1: 0}
    for v in vectors:
        if v["m"] * max(2, int(np.ceil(np.log2(v["sigma"])))) > 64 or v["n"] > 130000:
            continue
        text, pat = cases.build(v)
        for layout in (1, 0):
            knob.set(T.TUNE_KEY, "layout=%d" % layout)
            try:
                k = T.KeyTable(pat, v["m"], v["p"], v["sigma"])
            except T.SmhError:
                assert layout == 1  # not a set the bucket image takes
                continue
            assert k.info().layout == layout
            assert E.keys_scan(k, text, blocks=1) == v["count_ac"], (v["name"], layout)
            k.close()
            ran[layout] += 1
    assert ran[0] > 90 and ran[1] > 30, ran
