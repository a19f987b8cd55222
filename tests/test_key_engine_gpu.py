"""-m gpu: the key engine's kernels through the C ABI (smh_keys_*) against the oracle, the brute-force definition and the
reference's golden vectors.  Sizes the CPU checker finishes in seconds; full-size properties are in test_gpu_configs.py."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import emu_lib  # noqa: F401  (puts the package directory on sys.path)
import smatcher_hip as S
from test_key_engine import BUCKET_SETS, SETS, _text_and_patterns

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _count(k, text):
    dev = torch.device("cuda", 0)
    n = len(text)
    t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
    t[:n] = torch.from_numpy(np.ascontiguousarray(text)).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    k.scan_device(t.data_ptr(), n, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return int(cnt.item())


@pytest.mark.parametrize("sigma,m,p", SETS)
def test_kernel_counts_what_the_definition_counts(sigma, m, p):
    n = (1 << 20) + 4321
    text, pat = _text_and_patterns(sigma, m, p, n)
    k = S.KeyTable(pat, m, p, sigma)
    assert _count(k, text) == O.count_bruteforce(pat, m, p, text) > 0
    for cut in (0, 1, m - 1, m, 63, 4096, 4097, 8192 + 17):  # tails and texts shorter than a wave-chunk
        assert _count(k, text[:cut]) == O.count_bruteforce(pat, m, p, text[:cut]), cut
    k.close()


@pytest.mark.parametrize("sigma,m,p", BUCKET_SETS)
def test_both_images_of_a_set_count_the_same_on_the_device(sigma, m, p, knob):
    """round 6: the bucket image (one LDS read per column, overflow table behind a wave-uniform branch) and the cuckoo image of the same
    set, each forced in the testing twin, against the definition: 1 MiB with every second pattern cut from the text, ragged tails,
    positions"""
    T = knob.T
    n = (1 << 20) + 4321
    text, pat = _text_and_patterns(sigma, m, p, n)
    want = O.count_bruteforce(pat, m, p, text)
    for layout in (1, 0):
        knob.set(T.TUNE_KEY, "layout=%d" % layout)
        k = T.KeyTable(pat, m, p, sigma)
        assert k.info().layout == layout
        assert _count(k, text) == want > 0, layout
        for cut in (0, m - 1, m, 4096, 4097, 8192 + 17, 3 * 4096 + 5):
            assert _count(k, text[:cut]) == O.count_bruteforce(pat, m, p, text[:cut]), (layout, cut)
        dev = torch.device("cuda", 0)
        t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
        t[:n] = torch.from_numpy(np.ascontiguousarray(text)).to(dev)
        out = torch.zeros(want + 8, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        k.positions_device(t.data_ptr(), n, out.data_ptr(), want + 8, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == want
        assert np.array_equal(np.sort(out[:want].cpu().numpy()), np.sort(np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.int64))), layout
        k.close()


def test_golden_vectors_of_the_reference():
    """the counts the reference's own search_ac / search_wu2 produced (tests/golden/ref_vectors.json) for every vector the engine takes"""
    import cases
    vectors = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
    taken = 0
    for v in vectors:
        sigma, m, p = v["sigma"], v["m"], v["p"]
        if m * max(2, int(np.ceil(np.log2(sigma)))) > 64:
            continue
        text, pat = cases.build(v)
        k = S.KeyTable(pat, m, p, sigma)
        assert _count(k, text) == v["count_ac"], v["name"]
        k.close()
        taken += 1
    assert taken > 100


def test_positions_on_the_device():
    sigma, m, p, n = 4, 16, 500, (1 << 20) + 99
    text, pat = _text_and_patterns(sigma, m, p, n)
    want = np.asarray(O.positions_bruteforce(pat, m, p, text), dtype=np.int64)
    dev = torch.device("cuda", 0)
    t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
    t[:n] = torch.from_numpy(text).to(dev)
    out = torch.zeros(len(want) + 16, dtype=torch.int64, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    k = S.KeyTable(pat, m, p, sigma)
    k.positions_device(t.data_ptr(), n, out.data_ptr(), len(want) + 16, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == len(want)
    assert np.array_equal(np.sort(out[:len(want)].cpu().numpy()), np.sort(want))
    k.close()


def _scan_handle(h, text):
    dev = torch.device("cuda", 0)
    n = len(text)
    t = torch.zeros(((n + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
    t[:n] = torch.from_numpy(np.ascontiguousarray(text)).to(dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return int(cnt.item())


def test_engine_forced_inside_the_handles_on_the_golden_vectors():
    """smh_ac / smh_wm handles with SMH_ENGINE_KEYS forced: the reference's golden counts (every vector whose handle keeps a key table)"""
    import cases
    vectors = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
    forced = {"ac": 0, "wm": 0}
    for v in vectors:
        sigma, m, p = v["sigma"], v["m"], v["p"]
        text, pat = cases.build(v)
        for entry, make in (("ac", S.AcAutomaton), ("wm", S.WmTables)):
            if entry == "wm" and (m < 3 or sigma not in (2, 4, 8, 20, 128, 256)):
                continue
            h = make.from_patterns(pat, m, p, sigma)
            if h.info().key_slots:
                h.set_scan_engine(S.ENGINE_KEYS)
                assert _scan_handle(h, text) == v["count_ac"], (entry, v["name"])
                got, _ = h.count_host(text)  # the legacy host-pointer path through the same engine
                assert got == v["count_ac"], (entry, v["name"])
                forced[entry] += 1
            h.close()
    assert forced["ac"] >= 10 and forced["wm"] >= 10, forced


def test_handle_settles_on_the_key_table_on_hostile_text():
    """8000 patterns of 16 symbols sampled from the repeat-rich text: the filter kernels verify a real match in every tenth column,
    the plain stride-1 parts take four passes -- the key table one.  Same count from every engine the handle holds."""
    n, m, p, sigma = 256 << 20, 16, 8000, 4
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    S.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, S.CORPUS_DNA_REPEATS, st)
    torch.cuda.synchronize()
    pat = S.corpus_patterns(m, p, 12, sigma, 42, n, 2, S.CORPUS_DNA_REPEATS)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for make in (S.AcAutomaton, S.WmTables):
        h = make.from_patterns(pat, m, p, sigma)
        assert h.info().key_slots > 0 and h.info().adaptive == 1
        seen, counts = [], set()
        for _ in range(12):
            cnt.zero_()
            h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
            torch.cuda.synchronize()
            counts.add(int(cnt.item()))
            seen.append(int(h.adapt().engine))
        assert len(counts) == 1, counts
        assert seen[-1] == S.ENGINE_KEYS, seen
        for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT, S.ENGINE_KEYS):
            try:
                h.set_scan_engine(eng)
            except S.SmhError:
                continue
            cnt.zero_()
            h.scan_device(text.data_ptr(), 64 << 20, cnt.data_ptr(), S.VARIANT_TUNED, st)
            torch.cuda.synchronize()
            counts.add(("prefix", int(cnt.item())))
        assert len(counts) == 2, counts
        want = O.oracle_ac(pat, m, p, sigma, text[:8 << 20].cpu().numpy())[0]
        h.set_scan_engine(S.ENGINE_KEYS)
        cnt.zero_()
        h.scan_device(text.data_ptr(), 8 << 20, cnt.data_ptr(), S.VARIANT_TUNED, st)
        torch.cuda.synchronize()
        assert int(cnt.item()) == want
        h.close()
