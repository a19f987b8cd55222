"""Seeded test inputs shared by the golden-vector generator and the parity tests.

A case is a small dict; the text and patterns are regenerated from it with the
oracle's splitmix64 generator (the product's own generator is checked equal to
it in test_host_tables.py), so only seeds, sizes and expected results are
committed under tests/golden/.
"""
import numpy as np

import oracle_lib as O

SIGMAS = [2, 4, 8, 20, 128, 256]
LENGTHS = [3, 4, 8, 16, 32]
COUNTS = [1, 2, 100, 1000]


def matrix_cases():
    """SURVEY.md 8c fixture matrix: m x sigma x p, patterns half sampled from the text."""
    out = []
    k = 0
    for sigma in SIGMAS:
        for m in LENGTHS:
            for p in COUNTS:
                # odd sizes so that tails / partial segments are always exercised
                n = 60000 + 4099 * (k % 7) + (k % 16)
                out.append(dict(name="mx_s%d_m%d_p%d" % (sigma, m, p), n=n, p=p, m=m, sigma=sigma,
                                text="uniform", pat="mixed", text_seed=42 + k, pat_seed=7 + k))
                k += 1
    return out


def special_cases():
    out = []
    # SURVEY.md 8c known answers (uniform random patterns, text seed 42, pattern seed 7)
    out.append(dict(name="kat_1m_100x8", n=1 << 20, p=100, m=8, sigma=4, text="uniform", pat="uniform",
                    text_seed=42, pat_seed=7))
    out.append(dict(name="kat_1m003_r3", n=1000003, p=100, m=8, sigma=4, text="uniform", pat="uniform",
                    text_seed=42, pat_seed=7))
    # duplicates in the pattern set count once (ac/ac.c:183, wu/wu.c:91-95)
    out.append(dict(name="dups", n=70001, p=200, m=8, sigma=4, text="uniform", pat="dups",
                    text_seed=3, pat_seed=4))
    # overlapping matches: constant text, the all-zero pattern among others
    out.append(dict(name="overlap_zeros", n=50021, p=20, m=8, sigma=4, text="zeros", pat="with_zero",
                    text_seed=5, pat_seed=6))
    out.append(dict(name="overlap_zeros_m32", n=40009, p=20, m=32, sigma=2, text="zeros", pat="with_zero",
                    text_seed=5, pat_seed=6))
    # text shorter than / equal to / barely longer than the pattern
    out.append(dict(name="n_lt_m", n=5, p=10, m=8, sigma=4, text="uniform", pat="uniform", text_seed=8, pat_seed=9))
    out.append(dict(name="n_eq_m", n=8, p=10, m=8, sigma=4, text="uniform", pat="first_window",
                    text_seed=8, pat_seed=9))
    out.append(dict(name="n_eq_m_plus1", n=9, p=10, m=8, sigma=4, text="uniform", pat="first_window",
                    text_seed=8, pat_seed=9))
    # sizes around the kernels' geometry: 64-byte segments, 4 KiB / 8 KiB wave-chunks, 16-byte loads
    for n in (63, 64, 65, 4095, 4096, 4097, 4103, 8191, 8192, 8193, 8200, 8207, 8208, 12288 + 15, 16384 + 7,
              3 * 8192 + 16 + 6):
        out.append(dict(name="edge_n%d" % n, n=n, p=64, m=8, sigma=4, text="uniform", pat="mixed",
                        text_seed=100 + n, pat_seed=200 + n))
    for m in (17, 18, 33, 34, 64, 65, 66, 80):
        out.append(dict(name="edge_m%d" % m, n=40000 + m, p=50, m=m, sigma=4, text="uniform", pat="mixed",
                        text_seed=300 + m, pat_seed=400 + m))
    # dense hits: many short patterns on a small alphabet (every SHIFT entry 0)
    out.append(dict(name="dense_dna", n=120007, p=3000, m=8, sigma=4, text="uniform", pat="uniform",
                    text_seed=11, pat_seed=12))
    out.append(dict(name="dense_bin", n=90001, p=500, m=12, sigma=2, text="uniform", pat="uniform",
                    text_seed=13, pat_seed=14))
    # automaton too big for 16-bit rows (> 32768 kept states) and for the LDS budget
    out.append(dict(name="big_dfa", n=100003, p=2500, m=32, sigma=4, text="uniform", pat="mixed",
                    text_seed=15, pat_seed=16))
    out.append(dict(name="ascii_5_20", n=150001, p=3000, m=20, sigma=256, text="uniform", pat="mixed",
                    text_seed=17, pat_seed=18))
    out.append(dict(name="ascii_m5", n=150001, p=5000, m=5, sigma=256, text="uniform", pat="mixed",
                    text_seed=19, pat_seed=20))
    return out


def all_cases():
    return special_cases() + matrix_cases()


def build(case):
    """-> (text uint8[n], pat_flat uint8[p*m])"""
    n, p, m, sigma = case["n"], case["p"], case["m"], case["sigma"]
    if case["text"] == "uniform":
        text = O.gen_text(n, case["text_seed"], sigma)
    elif case["text"] == "zeros":
        text = np.zeros(n, dtype=np.uint8)
    else:
        raise ValueError(case["text"])
    kind = case["pat"]
    if kind == "uniform":
        pat = O.gen_patterns(m, p, case["pat_seed"], sigma)
    elif kind == "mixed":
        pat = O.gen_patterns_mixed(m, p, case["pat_seed"], sigma, case["text_seed"], n, 2)
    elif kind == "dups":
        pat = O.gen_patterns_mixed(m, p, case["pat_seed"], sigma, case["text_seed"], n, 2).reshape(p, m)
        pat[p // 2:] = pat[:p - p // 2]
        pat = pat.reshape(-1)
    elif kind == "with_zero":
        pat = O.gen_patterns(m, p, case["pat_seed"], sigma).reshape(p, m)
        pat[p // 3] = 0
        pat = pat.reshape(-1)
    elif kind == "first_window":
        pat = O.gen_patterns(m, p, case["pat_seed"], sigma).reshape(p, m)
        pat[p - 1] = text[:m]
        pat = pat.reshape(-1)
    else:
        raise ValueError(kind)
    return text, np.ascontiguousarray(pat, dtype=np.uint8)


# ------------------------------------------------------------------ mixed-length pattern sets
def mixed_cases():
    """Pattern sets with several lengths.  The reference takes one m per run, so the
    expected value is the length-class decomposition: the reference run once per distinct length, summed."""
    return [
        dict(name="mixed_dna_8_32", n=200003, sigma=4, classes=[[8, 200], [12, 200], [16, 200], [24, 200], [32, 200]],
             kind="classes", text_seed=61, pat_seed=62),
        dict(name="mixed_ascii_5_20", n=200017, sigma=256, classes=[[L, 150] for L in range(5, 21)],
             kind="classes", text_seed=63, pat_seed=64),
        dict(name="mixed_protein", n=150001, sigma=20, classes=[[3, 40], [7, 300], [9, 300]],
             kind="classes", text_seed=65, pat_seed=66),
        dict(name="mixed_len_1_2", n=90001, sigma=8, classes=[[1, 3], [2, 10], [3, 30], [5, 100]],
             kind="classes", text_seed=67, pat_seed=68),
        # short patterns that are prefixes of longer ones -- what ac_addstring cannot take in one trie
        # (ac/ac.c:136-143) -- plus duplicates inside a class
        dict(name="mixed_prefix_hazard", n=120007, sigma=4, classes=[[4, 50], [6, 100], [10, 100]],
             kind="prefixes", text_seed=69, pat_seed=70),
    ]


def build_mixed(case):
    """-> (text uint8[n], patterns uint8[sum lengths] back to back, lengths uint32[p])"""
    n, sigma = case["n"], case["sigma"]
    text = O.gen_text(n, case["text_seed"], sigma)
    per_class = {}
    for L, p in case["classes"]:
        per_class[L] = O.gen_patterns_mixed(L, p, case["pat_seed"] + L, sigma, case["text_seed"], n, 2).reshape(p, L)
    if case["kind"] == "prefixes":
        long = per_class[10]
        per_class[4] = np.ascontiguousarray(long[:50, :4])
        six = np.ascontiguousarray(long[:100, :6])
        six[50:] = six[:50]  # duplicates
        per_class[6] = six
    items = [(L, j) for L, p in case["classes"] for j in range(p)]
    order = np.random.RandomState(case["pat_seed"]).permutation(len(items))
    pats, lengths = [], []
    for k in order:
        L, j = items[k]
        pats.append(per_class[L][j])
        lengths.append(L)
    return text, np.ascontiguousarray(np.concatenate(pats), dtype=np.uint8), np.asarray(lengths, dtype=np.uint32)


def split_classes(patterns, lengths):
    """length -> flat uint8 array of that class's patterns, in the order given"""
    out, off = {}, 0
    for L in lengths:
        out.setdefault(int(L), []).append(patterns[off:off + int(L)])
        off += int(L)
    return {L: np.ascontiguousarray(np.concatenate(v), dtype=np.uint8) for L, v in out.items()}


def sog_cases():
    """SOG inputs (sog/sog8.c: patterns of length 8 over raw bytes): built with build()."""
    out = []
    for name, n, p, sigma, pat in [("sog_dna_100", 300007, 100, 4, "mixed"), ("sog_dna_1000", 500009, 1000, 4, "mixed"),
                                   ("sog_protein_500", 400003, 500, 20, "mixed"), ("sog_ascii_2000", 600011, 2000, 256, "mixed"),
                                   ("sog_dups", 200003, 64, 4, "dups"), ("sog_binary", 150001, 40, 2, "mixed"),
                                   ("sog_short_text", 11, 3, 4, "first_window"), ("sog_one", 70001, 1, 8, "mixed")]:
        out.append(dict(name=name, n=n, p=p, m=8, sigma=sigma, text="uniform", text_seed=4242, pat=pat, pat_seed=77))
    return out
