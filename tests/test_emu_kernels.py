"""The kernels' lane code (csrc/ac_lane.h, csrc/wm_lane.h), compiled for the CPU and driven thread
by thread over the launch grid, against the reference's golden counts.  This is the GPU-less
check of tiling / halo / tail / early-exit / hot-cold logic; the real kernels are checked through
the C ABI in test_gpu_parity.py (-m gpu)."""
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
    VECTORS = json.load(f)


@pytest.mark.parametrize("vec", VECTORS, ids=[v["name"] for v in VECTORS])
def test_emulated_kernels_match_reference_counts(vec):
    text, pat = cases.build(vec)
    p, m, sigma, want = vec["p"], vec["m"], vec["sigma"], vec["count_ac"]
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    rows = ac.info().rows
    assert E.ac_scan(ac, text, S.VARIANT_TUNED, 0, 3) == want           # whole DFA in "LDS"
    assert E.ac_scan(ac, text, S.VARIANT_TUNED, max(1, rows // 3), 1) == want  # hot/cold split
    assert E.ac_scan(ac, text, S.VARIANT_TABLE, 0, 2) == want           # goto/supply/final walk
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert E.wm_scan(wm, text, S.VARIANT_TUNED, 3) == want == vec["count_wu2"]
    assert E.wm_scan(wm, text, S.VARIANT_TABLE, 2) == want


def test_every_boundary_offset():
    """A single pattern occurrence slid across segment (64 B), wave-chunk (4/8 KiB) and text-end
    boundaries; the count must be exactly 1 (or 2 when planted twice) at every offset."""
    m, sigma = 8, 4
    pat = np.array([3, 1, 2, 0, 3, 3, 1, 2], dtype=np.uint8)
    ac = S.AcAutomaton.from_patterns(pat, m, 1, sigma)
    wm = S.WmTables.from_patterns(pat, m, 1, sigma)
    n = 8192 * 2 + 64 + 16 + 5
    offsets = list(range(0, 80)) + list(range(4096 - 70, 4096 + 10)) + list(range(8192 - 70, 8192 + 30)) + \
        list(range(16384 - 10, n - m + 1))
    for off in offsets:
        text = np.zeros(n, dtype=np.uint8)
        text[off:off + m] = pat
        for got in (E.ac_scan(ac, text, 0, 0, 1), E.ac_scan(ac, text, 1, 0, 1), E.wm_scan(wm, text, 0, 1),
                    E.wm_scan(wm, text, 1, 1)):
            assert got == 1, off


@pytest.mark.parametrize("m", [9, 24, 40, 70])
def test_long_patterns_straddling_segments(m):
    sigma = 4
    rng = np.random.RandomState(m)
    pat = rng.randint(0, sigma, size=m).astype(np.uint8)
    ac = S.AcAutomaton.from_patterns(pat, m, 1, sigma)
    wm = S.WmTables.from_patterns(pat, m, 1, sigma)
    n = 8192 + 4096 + 200
    for off in list(range(0, 70, 3)) + list(range(8192 - m - 2, 8192 + 3)) + [n - m]:
        text = rng.randint(0, sigma, size=n).astype(np.uint8)
        text[off:off + m] = pat
        want = O.count_bruteforce(pat, m, 1, text)
        assert want >= 1
        assert E.ac_scan(ac, text, 0, 0, 1) == want
        assert E.ac_scan(ac, text, 0, 5, 1) == want
        assert E.wm_scan(wm, text, 0, 1) == want
        assert E.wm_scan(wm, text, 1, 1) == want


def test_grid_size_does_not_change_the_count():
    vec = next(v for v in VECTORS if v["name"] == "dense_dna")
    text, pat = cases.build(vec)
    ac = S.AcAutomaton.from_patterns(pat, vec["m"], vec["p"], vec["sigma"])
    wm = S.WmTables.from_patterns(pat, vec["m"], vec["p"], vec["sigma"])
    for blocks in (1, 2, 5, 16):
        assert E.ac_scan(ac, text, 0, 0, blocks) == vec["count_ac"]
        assert E.wm_scan(wm, text, 0, blocks) == vec["count_ac"]
