"""Match-position output: END columns of all matches, compacted with a wave-level
prefix sum into a device buffer.  The reference only has commented-out printf's for positions
(ac/ac.c:217, wu/wu.c:93), so parity is the sorted list against the oracle's definition-level
brute force; the count of positions must also equal the reference's golden match count."""
import ctypes as C
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
    VECTORS = {v["name"]: v for v in json.load(f)}

NAMES = ["kat_1m_100x8", "dups", "overlap_zeros", "overlap_zeros_m32", "n_lt_m", "n_eq_m", "n_eq_m_plus1", "edge_n63",
         "edge_n4097", "edge_n8207", "edge_m33", "edge_m66", "dense_dna", "big_dfa", "ascii_m5", "mx_s20_m16_p100",
         "mx_s2_m32_p1000", "mx_s8_m3_p2"]


@pytest.mark.parametrize("name", NAMES)
def test_emulated_positions_match_bruteforce(name):
    vec = VECTORS[name]
    text, pat = cases.build(vec)
    m, p, sigma = vec["m"], vec["p"], vec["sigma"]
    want = O.positions_bruteforce(pat, m, p, text)
    assert len(want) == vec["count_ac"]
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    cap = len(want) + 5
    total, got = E.ac_positions(ac, text, cap, 3)
    assert total == len(want) and np.array_equal(np.sort(got).astype(np.int64), want)
    total, got = E.wm_positions(wm, text, cap, 2)
    assert total == len(want) and np.array_equal(np.sort(got).astype(np.int64), want)
    if len(want) > 3:
        # too small a buffer: the cursor still reports the full count, nothing is written past the end
        total, got = E.ac_positions(ac, text, 3, 1)
        assert total == len(want) and len(got) == 3 and set(got.astype(np.int64)) <= set(want.tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_positions_match_bruteforce(name):
    import torch
    vec = VECTORS[name]
    text, pat = cases.build(vec)
    m, p, sigma = vec["m"], vec["p"], vec["sigma"]
    want = O.positions_bruteforce(pat, m, p, text)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    dev = torch.device("cuda", 0)
    n = len(text)
    d_text = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_text[:n] = torch.from_numpy(text).to(dev)
    cap = len(want) + 7
    for obj in (ac, wm):
        out = torch.zeros(cap, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        obj.positions_device(d_text.data_ptr(), n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        total = int(cur.item())
        assert total == len(want) == vec["count_ac"]
        got = np.sort(out[:total].cpu().numpy())
        assert np.array_equal(got, want)
    if len(want) > 3:
        out = torch.full((8,), -1, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        ac.positions_device(d_text.data_ptr(), n, out.data_ptr(), 3, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == len(want) and (out[3:] == -1).all()
