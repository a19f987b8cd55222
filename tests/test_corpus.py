"""The synthetic corpora (csrc/corpus.c, csrc/corpus_gen.h): the uniform stream of SURVEY 8(c) and the round-4 kinds that
look like the reference's own data (main.c:39-109: genomes, proteins, English) -- every slice regenerable on its own, the
device generator byte-identical to the host's, and the properties each kind is there for."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402

KINDS = [(S.CORPUS_DNA_REPEATS, 4), (S.CORPUS_SKEWED, 20), (S.CORPUS_SKEWED, 256), (S.CORPUS_PLANTED, 4), (S.CORPUS_PLANTED, 256)]


@pytest.mark.parametrize("kind,sigma", KINDS)
def test_slices_are_regenerable_and_symbols_in_range(kind, sigma):
    whole = S.corpus_text(70_000, 42, sigma, 0, kind)
    assert whole.max() < sigma
    for off, n in ((0, 1), (1023, 2), (1024, 1024), (777, 5000), (65_000, 5000), (12_345, 54_321)):
        assert np.array_equal(S.corpus_text(n, 42, sigma, off, kind), whole[off:off + n])
    assert not np.array_equal(S.corpus_text(4096, 43, sigma, 0, kind), whole[:4096])  # the seed matters
    far = S.corpus_text(4096, 42, sigma, (1 << 33) + 5, kind)  # offsets beyond 2^32
    assert far.max() < sigma and len(np.unique(far)) > 1


def test_kind_zero_is_the_uniform_stream_and_bad_arguments_are_refused():
    assert np.array_equal(S.corpus_text(5000, 42, 4, 99, S.CORPUS_UNIFORM), S.corpus_text(5000, 42, 4, 99))
    with pytest.raises(S.SmhError):
        S.corpus_text(16, 42, 20, 0, S.CORPUS_DNA_REPEATS)  # a 4-letter text
    with pytest.raises(S.SmhError):
        S.corpus_text(16, 42, 4, 0, 9)
    with pytest.raises(S.SmhError):
        S.corpus_text(16, 42, 300, 0, S.CORPUS_SKEWED)


def test_dna_repeats_has_repeats_runs_and_markov_structure():
    n = 1 << 22
    t = S.corpus_text(n, 42, 4, 0, S.CORPUS_DNA_REPEATS)
    blocks = t.reshape(-1, 1024)
    # library blocks recur: many 1 KiB blocks are byte-identical to another one
    keys = {}
    for i, b in enumerate(blocks):
        keys.setdefault(b.tobytes(), []).append(i)
    copies = sum(len(v) for v in keys.values() if len(v) > 1)
    assert 0.10 < copies / len(blocks) < 0.22
    # low-complexity runs: blocks in which one symbol makes up > 90 %
    low = sum(1 for b in blocks if np.bincount(b, minlength=4).max() > 0.9 * 1024)
    assert 0.06 < low / len(blocks) < 0.15
    # tandem repeats: blocks that equal themselves shifted by a period of 2..31
    tandem = sum(1 for b in blocks[:1000] if np.bincount(b, minlength=4).max() <= 0.9 * 1024 and
                 any(np.array_equal(b[u:], b[:-u]) for u in range(2, 32)))
    assert 20 < tandem < 90
    # order-3 conditional entropy well below 2 bits
    ctx = (t[:-3].astype(np.int64) * 16 + t[1:-2] * 4 + t[2:-1]) * 4 + t[3:]
    c = np.bincount(ctx, minlength=256).reshape(64, 4).astype(float)
    p = c / c.sum(axis=1, keepdims=True)
    h3 = -(c / c.sum() * np.log2(np.where(p > 0, p, 1))).sum()
    assert h3 < 1.8


@pytest.mark.parametrize("sigma,lo,hi", [(20, 3.6, 4.3), (256, 3.8, 4.8)])
def test_skewed_frequencies(sigma, lo, hi):
    t = S.corpus_text(1 << 20, 42, sigma, 0, S.CORPUS_SKEWED)
    c = np.bincount(t, minlength=sigma) / len(t)
    h = -(c[c > 0] * np.log2(c[c > 0])).sum()
    assert lo < h < hi
    assert c[0] > 2.5 * c[min(sigma - 1, 15)]  # rank 0 far more frequent than rank 15


@pytest.mark.parametrize("sigma", [4, 256])
def test_planted_word_recurs_in_every_cell_and_is_pattern_zero(sigma):
    n, m = 1 << 16, 16
    t = S.corpus_text(n, 42, sigma, 0, S.CORPUS_PLANTED)
    pat = S.corpus_patterns(m, 10, 7, sigma, 42, n, 2, S.CORPUS_PLANTED).reshape(10, m)
    word = pat[0].tobytes()
    for cell in range(0, n, 64):
        assert word in t[cell:cell + 64].tobytes()
    word32 = S.corpus_patterns(32, 1, 7, sigma, 42, n, 0, S.CORPUS_PLANTED).tobytes()
    assert word32[:m] == word


@pytest.mark.parametrize("kind,sigma", KINDS)
def test_patterns_from_text_are_substrings(kind, sigma):
    n, m, p = 200_000, 12, 40
    t = S.corpus_text(n, 42, sigma, 0, kind).tobytes()
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, kind).reshape(p, m)
    assert pat.max() < sigma
    for j in range(0, p, 2):
        assert pat[j].tobytes() in t


@pytest.mark.gpu
@pytest.mark.parametrize("kind,sigma", KINDS)
def test_device_generator_equals_host(kind, sigma):
    import torch
    dev = torch.device("cuda", 0)
    for off, n in ((0, 1 << 20), (1024 * 5 + 16, 300_000), (777, 70_001), (1 << 32, 1 << 18)):
        t = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(t.data_ptr(), n, 42, sigma, off, kind)
        torch.cuda.synchronize()
        assert np.array_equal(t[:n].cpu().numpy(), S.corpus_text(n, 42, sigma, off, kind)), (off, n)
        assert int(t[n:].sum()) == 0  # nothing written past the slice
