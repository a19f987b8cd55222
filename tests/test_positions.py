"""Match-position output: END columns of all matches, compacted with a wave-level
prefix sum into a device buffer.  Two implementations behind smh_ac_positions / smh_wm_positions: the
tuned scan kernels in positions mode (matches recorded as bits / verified through the candidate queues,
appended per wave) and the per-segment kernels over the HBM tables (unaligned text, halos beyond 32
bytes); both are checked.  The reference only has commented-out printf's for positions
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
    # the tuned kernels in positions mode: every scan-table format, K = m and K < m
    total, got = E.wm_positions_tuned(wm, text, cap, 2)
    assert total == len(want) and np.array_equal(np.sort(got).astype(np.int64), want)
    plans = [(0, 0), (1, min(m, 33)), (1, max(1, m // 2)), (1, 2)]
    if sigma == 4:
        plans += [(2, min(m, 33)), (2, max(1, m // 2)), (3, min(m, 33) | (1 << 8)), (3, max(4, m // 2) | (2 << 8))]
    tried = 0
    for stride, depth in plans:
        try:
            ac.set_scan_plan(stride, depth)
        except S.SmhError:
            continue
        total, got = E.ac_positions_tuned(ac, text, cap, 2)
        if total is None:
            continue  # halo beyond 32 bytes: the runtime uses the per-segment kernel
        tried += 1
        assert total == len(want) and np.array_equal(np.sort(got).astype(np.int64), want), (stride, depth)
    assert tried >= 1 or m > 33


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
    if wm.info().scan_engine == S.ALGO_AC:  # the Wu-Manber path's own kernels in positions mode
        wm.set_scan_engine(S.ALGO_WM)
        out = torch.zeros(cap, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        wm.positions_device(d_text.data_ptr(), n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == len(want) and np.array_equal(np.sort(out[:len(want)].cpu().numpy()), want)
        wm.set_scan_engine(-1)
    # the per-segment kernels (what unaligned text gets): same positions, shifted by the 4-byte offset
    d_un = torch.zeros(n + 68, dtype=torch.uint8, device=dev)
    d_un[4:4 + n] = torch.from_numpy(text).to(dev)
    for obj in (ac, wm):
        out = torch.zeros(cap, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        obj.positions_device(d_un.data_ptr() + 4, n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == len(want) and np.array_equal(np.sort(out[:len(want)].cpu().numpy()), want)
    # other scan plans of the automaton in positions mode
    if sigma == 4 and m >= 6:
        for stride, depth in ((1, min(m, 33)), (2, min(m, 33)), (3, min(m, 33) | (1 << 8)), (1, max(2, m // 2)), (2, max(2, m // 2))):
            try:
                ac.set_scan_plan(stride, depth)
            except S.SmhError:
                continue
            out = torch.zeros(cap, dtype=torch.int64, device=dev)
            cur = torch.zeros(1, dtype=torch.int64, device=dev)
            ac.positions_device(d_text.data_ptr(), n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert int(cur.item()) == len(want) and np.array_equal(np.sort(out[:len(want)].cpu().numpy()), want), (stride, depth)
        ac.set_scan_plan(0, 0)
    if len(want) > 3:
        out = torch.full((8,), -1, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        ac.positions_device(d_text.data_ptr(), n, out.data_ptr(), 3, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == len(want) and (out[3:] == -1).all()
