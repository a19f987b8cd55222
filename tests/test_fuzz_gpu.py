"""Differential fuzzing of every engine, forced scan plan and positions path against the oracle's brute
force (tests/fuzz_gpu.py): pattern sets with long shared prefixes / suffixes, duplicates and text-cut
patterns over random alphabets, lengths and sizes.  `python tests/fuzz_gpu.py 80 <seed>` runs more."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_sets_agree_with_the_oracle(seed):
    import fuzz_gpu
    assert fuzz_gpu.run(25, seed, verbose=False) > 200


@pytest.mark.parametrize("m,p", [(12, 3000), (16, 1000), (16, 8000), (24, 8000), (32, 1000), (33, 300)])
def test_long_texts_with_every_verify_mode_and_engine_forced(m, p, knob):
    """Texts of several MiB -- many consecutive fast chunks per wave, pending columns carried from chunk to chunk -- with the
    pair-gram kernels' verify forced in registers, staged, and staged with the drain from HBM, and with every engine an
    Aho-Corasick handle keeps forced in turn (the automaton kernels, the filter kernels, the plain stride-1 automaton); a
    stretch of back-to-back repeats of one pattern and a poly-symbol run give some chunks hundreds of surviving columns."""
    import numpy as np
    import oracle_lib as O
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    sigma = 4
    n = (7 << 20) + 12345 + 64 * m
    text = S.corpus_text(n, 1000 + m, sigma)
    pat = S.corpus_patterns(m, p, 7 + m, sigma, 1000 + m, n, 2)
    text[3 << 20:(3 << 20) + 40 * m] = np.tile(pat[:m], 40)
    text[(5 << 20) - 3000:(5 << 20) + 3000] = pat[m]
    pat[2 * m:3 * m] = pat[m]  # a one-symbol pattern: every column of the run matches
    want = O.oracle_ac(pat, m, p, sigma, text)[0]
    assert want > 6000
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    wm.set_scan_engine(S.ALGO_WM)
    for tune in ("", "regv=1", "regv=0", "regv=0,hd=1", "regv=0,hd=0", "stage=0"):
        knob.wm(tune)
        assert wm.count_host(text)[0] == want, tune
    knob.wm(None)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    assert ac.count_host(text)[0] == want
    forced = 0
    for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT):
        try:
            ac.set_scan_engine(eng)
        except S.SmhError:
            continue
        forced += 1
        assert ac.count_host(text)[0] == want, eng
    assert forced >= 1
    S.lib.smh_host_path_release()
