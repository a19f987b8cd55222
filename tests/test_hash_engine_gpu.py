"""-m gpu: the window-hash engine's kernel (csrc/hash_kernels.hip) forced inside Wu-Manber handles, against the brute-force
definition, the oracle and the reference's golden vectors; and the handle settling on it on natural-language-like text."""
import json
import os

import numpy as np
import pytest
import torch

import emu_lib  # noqa: F401  (sys.path)
import oracle_lib as O
import smatcher_hip as S
from test_hash_engine import SETS, _case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _scan(h, text, n=None):
    dev = torch.device("cuda", 0)
    n = len(text) if n is None else n
    if isinstance(text, torch.Tensor):
        t = text
    else:
        t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
        t[:n] = torch.from_numpy(np.ascontiguousarray(text[:n])).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return int(cnt.item())


@pytest.mark.parametrize("sigma,m,p", SETS)
def test_kernel_counts_what_the_definition_counts(sigma, m, p):
    n = (1 << 20) + 4321
    text, pat = _case(sigma, m, p, n)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    wm.set_scan_engine(S.ENGINE_HASH)
    assert _scan(wm, text) == O.count_bruteforce(pat, m, p, text) > 0
    for cut in (0, 1, m - 1, m, 63, 4096, 4097, 4160, 4161, 8192 + 17):
        assert _scan(wm, text[:cut]) == O.count_bruteforce(pat, m, p, text[:cut]), cut
    got, _ = wm.count_host(text)
    assert got == O.count_bruteforce(pat, m, p, text)
    wm.close()


def test_golden_vectors_of_the_reference():
    import cases
    vectors = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
    taken = 0
    for v in vectors:
        if v["sigma"] not in (8, 20, 128, 256) or v["m"] < 4:
            continue
        text, pat = cases.build(v)
        wm = S.WmTables.from_patterns(pat, v["m"], v["p"], v["sigma"])
        if wm.info().hash_slots:
            wm.set_scan_engine(S.ENGINE_HASH)
            assert _scan(wm, text) == v["count_wu"], v["name"]
            taken += 1
        wm.close()
        if v["p"] <= 3000:  # the Aho-Corasick entry point runs the same engine (its filter engine's)
            ac = S.AcAutomaton.from_patterns(pat, v["m"], v["p"], v["sigma"])
            if ac.info().hash_slots:
                ac.set_scan_engine(S.ENGINE_HASH)
                assert _scan(ac, text) == v["count_ac"], v["name"]
                taken += 1
            ac.close()
    assert taken >= 16, taken


def test_positions_on_the_device():
    sigma, m, p, n = 256, 12, 500, (1 << 20) + 99
    text, pat = _case(sigma, m, p, n)
    want = np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.int64)
    dev = torch.device("cuda", 0)
    t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
    t[:n] = torch.from_numpy(text).to(dev)
    out = torch.zeros(len(want) + 16, dtype=torch.int64, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    wm.set_scan_engine(S.ENGINE_HASH)
    wm.positions_device(t.data_ptr(), n, out.data_ptr(), len(want) + 16, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == len(want)
    assert np.array_equal(np.sort(out[:len(want)].cpu().numpy()), np.sort(want))
    wm.close()


def test_handle_settles_on_it_on_natural_language_like_text():
    """100 000 patterns of 12 bytes sampled from the skewed 256-symbol corpus: the byte-gram filter passes half the columns of that
    text (its common grams are the patterns'), the window-hash filter the matches and ~4 %.  On uniform text the gram filter stays."""
    n, m, p, sigma = 256 << 20, 12, 100000, 256
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    pat = S.corpus_patterns(m, p, 12, sigma, 42, n, 2, S.CORPUS_SKEWED)
    for kind, want_engine in ((S.CORPUS_SKEWED, S.ENGINE_HASH), (S.CORPUS_UNIFORM, S.ALGO_WM)):
        text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
        torch.cuda.synchronize()
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        assert wm.info().hash_slots > 0 and wm.info().adaptive == 1
        counts, seen = set(), []
        for _ in range(10):
            counts.add(_scan(wm, text, n))
            seen.append(int(wm.adapt().engine))
        assert len(counts) == 1, counts
        assert seen[-1] == want_engine, seen
        # every engine the handle holds agrees on a prefix, and that prefix agrees with the restated search_wu2
        chk = 4 << 20
        pre = set()
        for eng in (S.ALGO_WM, S.ENGINE_HASH):
            wm.set_scan_engine(eng)
            pre.add(_scan(wm, text, chk))
        assert pre == {O.oracle_wu(pat, m, p, sigma, text[:chk].cpu().numpy())[0]}
        wm.close()
        del text


def test_cuckoo_form_of_the_verify_table_on_the_device(knob):
    """The filter kernels' pipelined probes read the verify entries from the two-table cuckoo hash or from the bucket table
    (csrc/wm_lane.h smh_wm_ck_*): 1.06x instead of 1.5-1.95x HBM traffic.  Round 6: the default of the 143.9 KiB filter forms, where
    the two run level; the testing twin's "ck=0|1" forces either -- both, and the default, give the reference's count."""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    sigma, p, n = 256, 40000, (4 << 20) + 77
    for m in (5, 8, 12, 20):
        pat = O.gen_patterns(m, p, 7, sigma)
        text = O.gen_text(n, 9, sigma)
        text[100000:100000 + m * 2000] = pat[:m * 2000]  # 2000 patterns back to back: true matches through the cuckoo probe
        want = O.oracle_wu(pat, m, p, sigma, text)[0]
        assert want >= 2000
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        assert wm.info().verify_ck_slots > 0
        wm.set_scan_engine(S.ALGO_WM)
        for tune in ("ck=1", "ck=0", None):
            knob.wm(tune)
            assert _scan(wm, text) == want, (m, tune)
        wm.close()
