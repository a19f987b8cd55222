"""The legacy host-pointer path (search_ac / search_wu / cuda_* and the *_count_host helpers): the text crosses PCIe in
pieces through two device buffers of a pooled workspace, piece k+1 copying while piece k is scanned, pieces overlapping by
m - 1 bytes (main.c:467-477).  smh_host_path_set_piece shrinks the pieces so that a small text has hundreds of boundaries."""
import os
import sys

import numpy as np
import pytest

from perf import perf_check

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402


def test_release_entry_point_exists_and_is_harmless_without_a_device():
    S.lib.smh_host_path_release()
    S.lib.smh_host_path_release()


def test_piece_size_setter_rounds_and_restores():
    default = S.lib.smh_host_path_set_piece(0)
    assert default >= 4096 and default % 4096 == 0
    assert S.lib.smh_host_path_set_piece(5000) == default
    assert S.lib.smh_host_path_set_piece(1) == 4096      # 5000 rounded down to 4 KiB
    assert S.lib.smh_host_path_set_piece(0) == 4096      # 1 raised to the 4 KiB floor
    assert S.lib.smh_host_path_set_piece(0) == default


@pytest.fixture
def piece():
    yield lambda kib: S.lib.smh_host_path_set_piece(kib << 10)
    S.lib.smh_host_path_set_piece(0)


@pytest.mark.gpu
@pytest.mark.parametrize("piece_kib", [4, 8, 64, 0])
@pytest.mark.parametrize("m,p,sigma", [(8, 300, 4), (16, 500, 4), (33, 200, 4), (5, 2000, 256), (12, 3000, 256), (65, 20, 4)])
def test_counts_across_piece_boundaries(piece, piece_kib, m, p, sigma):
    piece(piece_kib)  # 0: the default, 64 MiB
    for n in (1_000_003, 4096 * 7, 4096 * 7 + m - 1, 4096 * 7 + m - 2, 4095, m, m - 1, 0):
        text = S.corpus_text(n, 42, sigma)
        pat = S.corpus_patterns(m, p, 7, sigma, 42, max(n, m), 2)
        # a match that straddles every kind of boundary: copies of pattern 0 across the piece edges
        if n > 3 * 4096 + m:
            for edge in (4096, 8192, 12288):
                text[edge - m // 2:edge - m // 2 + m] = pat[:m]
                text[edge - 1:edge - 1 + m] = pat[m:2 * m]
                text[edge - m + 1:edge + 1] = pat[2 * m:3 * m]
        want = O.oracle_ac(pat, m, p, sigma, text)[0] if n >= m else 0
        ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
        for variant in (S.VARIANT_TUNED, S.VARIANT_TABLE):
            got, secs = ac.count_host(text, variant)
            assert got == want, (n, variant, got, want)
            assert secs >= 0
        if m >= 3:
            wm = S.WmTables.from_patterns(pat, m, p, sigma)
            for variant in (S.VARIANT_TUNED, S.VARIANT_TABLE):
                assert wm.count_host(text, variant)[0] == want, (n, variant)
            wm.close()
        ac.close()
    S.lib.smh_host_path_release()


@pytest.mark.gpu
def test_siblings_and_legacy_names_through_the_pieces(piece):
    piece(4)
    n, m, p, sigma = 300_001, 8, 200, 4
    text = S.corpus_text(n, 42, sigma)
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    want = O.oracle_ac(pat, m, p, sigma, text)[0]
    for cls in (S.ShTrie, S.SbomOracle):
        h = cls.from_patterns(pat, m, p, sigma)
        for variant in (S.VARIANT_TUNED, S.VARIANT_TABLE):
            assert h.count_host(text, variant=variant)[0] == want
        h.close()
    sog = S.SogTables(pat, p)
    for variant in (S.VARIANT_TUNED, S.VARIANT_TABLE):
        assert sog.count_host(text, variant)[0] == want
    sog.close()
    # the reference's own call shape: preproc_ac / search_ac / free_ac on caller-owned tables (main.c:138-153)
    C = S.C
    rows = [np.ascontiguousarray(np.append(r, 0).astype(np.uint8)) for r in pat.reshape(p, m)]
    ptrs = (S.u8p * p)(*[r.ctypes.data_as(S.u8p) for r in rows])
    st = np.full((m * p + 1) * sigma, -1, dtype=np.int32)
    su = np.zeros(m * p + 1, dtype=np.uint32)
    fi = np.zeros(m * p + 1, dtype=np.uint32)
    tab = S.lib.preproc_ac(ptrs, m, p, sigma, st.ctypes.data_as(S.i32p), su.ctypes.data_as(S.u32p), fi.ctypes.data_as(S.u32p))
    assert S.lib.search_ac(text.ctypes.data_as(S.u8p), n, tab) == want
    # the caller may rewrite its buffer between calls: nothing is cached by pointer
    text[:] = S.corpus_text(n, 43, sigma)
    assert S.lib.search_ac(text.ctypes.data_as(S.u8p), n, tab) == O.oracle_ac(pat, m, p, sigma, text)[0]
    S.lib.free_ac(tab, sigma)
    S.lib.smh_host_path_release()


@pytest.mark.gpu
@pytest.mark.perf
def test_host_path_rate_and_flat_device_memory():
    """1 GiB through the pieces at PCIe speed, with no device allocation per call after the first"""
    import time
    import torch
    n = 1 << 30
    text = S.corpus_text(n, 42, 4)
    pat = S.corpus_patterns(16, 1000, 7, 4, 42, n, 2)
    ac = S.AcAutomaton.from_patterns(pat, 16, 1000, 4)
    first = ac.count_host(text)[0]
    free0, _ = torch.cuda.mem_get_info()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        got, ksecs = ac.count_host(text)
        best = min(best, time.perf_counter() - t0)
        assert got == first and ksecs > 0
        perf_check(ksecs < 0.01, "smh_ac_count_host kernel_seconds %.4f for 1 GiB (expected < 0.01)" % ksecs)
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < (8 << 20)
    perf_check(n / best / 1e9 > 30.0, "host-pointer path at %.1f GB/s (expected > 30; ~55 on the round's boxes)" % (n / best / 1e9))
    S.lib.smh_host_path_release()
    assert torch.cuda.mem_get_info()[0] - free1 > (100 << 20)  # the two 64 MiB piece buffers went back
    ac.close()


@pytest.mark.gpu
def test_engine_flips_between_pieces_of_one_host_call():
    """A hostile text through the host-pointer path: every 64 MiB piece is a launch that reports, so the engine changes in the
    MIDDLE of one smh_ac_count_host call (the compile's choice for the first pieces, the plain stride-1 parts after): the count
    is the forced engines' and the oracle's on a slice."""
    n, m, p, sigma = 512 << 20, 32, 1000, 4
    text = S.corpus_text(n, 42, sigma, 0, S.CORPUS_PLANTED)
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, S.CORPUS_PLANTED)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    assert ac.info().adaptive == 1 and ac.info().flat_parts >= 2
    first = ac.info().scan_engine
    got = ac.count_host(text)[0]
    ad = ac.adapt()
    assert ad.flips >= 1 and ad.engine in (S.ENGINE_AC_FLAT, S.ENGINE_KEYS) and ad.reports >= 3, (ad.flips, ad.engine, ad.reports)
    perf_check(ad.ms_per_gib[first] > 3.0 * ad.est_ms_per_gib[first], "the compile's engine on the hostile text: %.3f ms/GiB measured, %.3f estimated (expected > 3x)" % (ad.ms_per_gib[first], ad.est_ms_per_gib[first]))
    ac.set_scan_engine(S.ENGINE_AC_FLAT)
    assert ac.count_host(text)[0] == got
    ac.set_scan_engine(first)
    assert ac.count_host(text[:96 << 20])[0] == O.oracle_ac(pat, m, p, sigma, text[:96 << 20])[0]
    ac.close()
    S.lib.smh_host_path_release()
