"""-m gpu: the key engine's kernels through the C ABI (smh_keys_*) against the oracle, the brute-force definition and the
reference's golden vectors.  Sizes the CPU checker finishes in seconds; full-size properties are in test_gpu_configs.py."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import smatcher_hip as S
from test_key_engine import SETS, _text_and_patterns

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
