"""Set Backward Oracle Matching (sbom/sbom.c, cuda/cuda_sbom.cu) -- the second
sibling algorithm behind the same API.  Expected values come from the reference's own compiled
sbom/sbom.c (tests/golden/ref_vectors.json: count_sbom, sbom_idcounter, sbom_patterncounter, digests of
state_transition -- trie edges and external transitions -- and of the state_final_multi rows).
CPU: the oracle equals the reference; preproc_sbom fills the caller's tables bit-identically; the lane
code (the reference's loop over the tables, and the tuned engine) reproduces every count.
GPU: search_sbom / cuda_sbom1..5 / smh_sbom_scan on the device."""
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
    VECTORS = json.load(f)
BY_NAME = {v["name"]: v for v in VECTORS}
IDS = [v["name"] for v in VECTORS]


def legacy_preproc(pat, m, p, sigma):
    """preproc_sbom through the legacy C name, tables allocated as main.c:410-425 does"""
    t = O.SBOMTables(m, p, sigma)
    rows = np.zeros((p, m + 1), dtype=np.uint8)
    rows[:, :m] = pat.reshape(p, m)
    arr = (S.u8p * p)()
    for j in range(p):
        arr[j] = C.cast(rows[j].ctypes.data, S.u8p)
    tab = S.lib.preproc_sbom(arr, m, p, sigma, t.state_transition.ctypes.data_as(S.i32p),
                             t.state_final_multi.ctypes.data_as(S.u32p))
    t.idcounter, t.patterncounter = tab.contents.idcounter, tab.contents.patterncounter
    return t, tab, (arr, rows)


@pytest.mark.parametrize("vec", VECTORS, ids=IDS)
def test_oracle_and_host_tables_match_the_reference(vec):
    text, pat = cases.build(vec)
    m, p, sigma = vec["m"], vec["p"], vec["sigma"]
    count, t = O.oracle_sbom(pat, m, p, sigma, text)
    assert count == vec["count_sbom"] == vec["count_ac"]  # same quantity as search_ac / search_wu / search_sh
    host, tab, keep = legacy_preproc(pat, m, p, sigma)
    for tabs in (t, host):
        assert (tabs.idcounter, tabs.patterncounter) == (vec["sbom_idcounter"], vec["sbom_patterncounter"])
        assert "%016x" % O.fnv(tabs.state_transition[:tabs.idcounter * sigma]) == vec["fnv_sbom_transition"]
        assert "%016x" % O.fnv(tabs.state_final_multi[:tabs.idcounter * 200]) == vec["fnv_sbom_final"]
    S.lib.free_sbom(tab, m)


@pytest.mark.parametrize("vec", VECTORS, ids=IDS)
def test_emulated_lane_code_matches_reference_counts(vec):
    text, pat = cases.build(vec)
    m, p, sigma, want = vec["m"], vec["p"], vec["sigma"], vec["count_sbom"]
    t = O.oracle_sbom(pat, m, p, sigma)[1]
    sb = S.SbomOracle.from_tables(pat, m, p, sigma, t.state_transition, t.state_final_multi, m * p + 1)
    info = sb.info()
    assert (info.states, info.patterns, info.listed) == (vec["sbom_idcounter"], p, p)
    assert E.sbom_scan(sb, text, S.VARIANT_TABLE, 2) == want   # the reference's loop over the tables
    assert E.sbom_scan(sb, text, S.VARIANT_TUNED, 2) == want   # tuned engine
    assert E.sbom_scan(S.SbomOracle.from_patterns(pat, m, p, sigma), text, S.VARIANT_TABLE, 1) == want


def test_bad_tables_are_reported():
    vec = BY_NAME["kat_1m_100x8"]
    _, pat = cases.build(vec)
    m, p, sigma = vec["m"], vec["p"], vec["sigma"]
    t = O.oracle_sbom(pat, m, p, sigma)[1]
    bad = t.state_final_multi.copy()
    bad[5 * 200] = 1
    bad[5 * 200 + 1] = p + 3  # a pattern id that does not exist
    t.state_transition[0] = 5
    with pytest.raises(S.SmhError, match="lists pattern"):
        S.SbomOracle.from_tables(pat, m, p, sigma, t.state_transition, bad, m * p + 1)
    t.state_transition[1] = m * p + 7  # an edge out of the table
    with pytest.raises(S.SmhError, match="leaves the table"):
        S.SbomOracle.from_tables(pat, m, p, sigma, t.state_transition, t.state_final_multi, m * p + 1)
    # the reference's 200-entry rows: 199 patterns per oracle state; the handle API reports it, it does not exit
    with pytest.raises(S.SmhError, match="199 patterns"):
        S.SbomOracle.from_patterns(np.tile(np.array([0, 1, 2, 3], dtype=np.uint8), 250), 4, 250, 4)
    assert S.SbomOracle.from_patterns(pat, m, p, sigma).info().tuned_engine == S.ALGO_WM
    assert S.SbomOracle.from_patterns(np.array([0, 1, 1, 0], dtype=np.uint8), 2, 2, 4).info().tuned_engine == S.ALGO_AC
    if S.device_count() == 0:
        with pytest.raises(S.SmhError):
            S.SbomOracle.from_patterns(pat, m, p, sigma).count_host(np.zeros(100, dtype=np.uint8))  # no CPU fallback


GPU_NAMES = ["kat_1m_100x8", "dups", "overlap_zeros", "overlap_zeros_m32", "n_lt_m", "n_eq_m", "edge_n4097", "edge_m33",
             "edge_m65", "dense_dna", "big_dfa", "ascii_5_20", "ascii_m5", "mx_s20_m16_p100", "mx_s2_m32_p1000",
             "mx_s8_m3_p2", "mx_s128_m8_p1000", "mx_s256_m4_p100", "mx_s2_m3_p1000"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_NAMES)
def test_gpu_sbom_counts(name, capfd):
    vec = BY_NAME[name]
    text, pat = cases.build(vec)
    n, m, p, sigma, want = vec["n"], vec["m"], vec["p"], vec["sigma"], vec["count_sbom"]
    tp = text.ctypes.data_as(S.u8p)
    t, tab, (arr, rows) = legacy_preproc(pat, m, p, sigma)
    assert S.lib.search_sbom(arr, m, tp, n, tab) == want  # multisbom, main.c:197-231
    S.lib.free_sbom(tab, m)
    capfd.readouterr()
    for k in range(1, 6):
        getattr(S.lib, "cuda_sbom%d" % k)(pat.ctypes.data_as(S.u8p), m, tp, n, p, sigma,
                                          t.state_transition.ctypes.data_as(S.i32p),
                                          t.state_final_multi.ctypes.data_as(S.u32p))
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("Kernel")]
    assert len(lines) == 5
    for k, ln in enumerate(lines, 1):
        parts = ln.split("\t")  # cuda/cuda_sbom.cu:212
        assert parts[0] == "Kernel %d matches " % k and int(parts[1]) == want
    sb = S.SbomOracle.from_patterns(pat, m, p, sigma)
    assert sb.count_host(text, S.VARIANT_TABLE)[0] == want == sb.count_host(text, S.VARIANT_TUNED)[0]
