"""The adaptive engine (csrc/smh_runtime.hip, csrc/smh_stats.h; round 4): a handle that holds several engines follows what
its launches report about the TEXT.  The compile chooses from rates measured on pseudo-random text; the reference's own
corpora (main.c:39-109) are genomes, proteins and English.  Counts never depend on the engine."""
import os
import sys

import numpy as np
import pytest

from perf import perf_check

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402


def test_adapt_entry_points_exist_and_report_without_a_device():
    pat = S.corpus_patterns(16, 300, 7, 4, 42, 1 << 20, 2)
    ac = S.AcAutomaton.from_patterns(pat, 16, 300, 4)
    info, ad = ac.info(), ac.adapt()
    assert info.adaptive == ad.adaptive
    assert ad.reports == 0 and ad.flips == 0 and ad.engine == info.scan_engine
    assert ad.est_ms_per_gib[S.ALGO_AC] > 0
    if info.adaptive:  # a depth-cut or hybrid plan keeps a second engine
        assert ad.est_ms_per_gib[S.ALGO_WM] > 0 or ad.est_ms_per_gib[S.ENGINE_AC_FLAT] > 0
    ac.set_scan_engine(S.ALGO_AC)  # a forced engine is never overridden
    assert ac.adapt().adaptive == 0 and ac.info().scan_engine == S.ALGO_AC
    ac.set_scan_engine(-1)
    assert ac.adapt().adaptive == info.adaptive
    bad = S.AdaptInfo(struct_size=8)
    assert S.lib.smh_ac_get_adapt(ac.h, S.C.byref(bad)) != 0
    wm = S.WmTables.from_patterns(pat, 16, 300, 4)
    assert wm.adapt().adaptive == wm.info().adaptive
    wm.close()
    ac.close()
    # the exact plans keep no second engine: their rate does not depend on the text
    pat8 = S.corpus_patterns(8, 300, 7, 4, 42, 1 << 20, 2)
    ac8 = S.AcAutomaton.from_patterns(pat8, 8, 300, 4)
    assert ac8.info().adaptive == 0
    with pytest.raises(S.SmhError):
        ac8.set_scan_engine(S.ENGINE_AC_FLAT)
    ac8.close()


def _dev_text(n, sigma, kind, seed=42):
    import torch
    t = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    S.corpus_text_device(t.data_ptr(), n, seed, sigma, 0, kind)
    torch.cuda.synchronize()
    return t


def _scan(h, t, n):
    import torch
    c = torch.zeros(1, dtype=torch.int64, device="cuda")
    h.scan_device(t.data_ptr(), n, c.data_ptr())
    torch.cuda.synchronize()
    return int(c.item())


@pytest.mark.gpu
@pytest.mark.perf
@pytest.mark.parametrize("entry", ["ac", "wm"])
def test_engine_follows_the_text_and_counts_do_not_change(entry):
    """1000 patterns of 16 symbols sampled from a repeat-rich genome-like text.  On uniform text the compile's choice (the
    pair-gram filter or the hybrid automaton) stands; on the repeat-rich text, where most lanes of the hybrid image sit in
    compact rows and every filter survivor is a real match, the handle moves to the plain stride-1 automaton within a few
    launches and runs several times faster than either forced engine; back on uniform text it returns."""
    n, m, p, sigma = 256 << 20, 16, 1000, 4
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, S.CORPUS_DNA_REPEATS)
    h = (S.AcAutomaton if entry == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
    assert h.info().adaptive == 1
    uni, rep = _dev_text(n, sigma, S.CORPUS_UNIFORM), _dev_text(n, sigma, S.CORPUS_DNA_REPEATS)
    host = {"uni": uni[:n].cpu().numpy(), "rep": rep[:n].cpu().numpy()}
    want = {k: O.oracle_ac(pat, m, p, sigma, v[:8 << 20])[0] for k, v in host.items()}
    assert _scan(h, uni, 8 << 20) == want["uni"] and _scan(h, rep, 8 << 20) == want["rep"]  # small launches do not report
    assert h.adapt().reports == 0
    first = h.info().scan_engine
    counts = {k: None for k in host}

    def run(text, key, launches):
        engines = []
        for _ in range(launches):
            c = _scan(h, text, n)
            counts[key] = c if counts[key] is None else counts[key]
            assert c == counts[key]  # whichever engine ran
            engines.append(int(h.adapt().engine))
        return engines

    e_uni = run(uni, "uni", 6)
    assert e_uni[0] == first and h.adapt().reports >= 2  # launches report until the running engine has two on record, then every eighth
    calm = h.adapt()
    assert calm.ms_per_gib[calm.engine] > 0, e_uni
    perf_check(calm.flips == 0 and calm.engine == first, "uniform text: %d flips, engine %d (compiled: %d): %r" % (calm.flips, calm.engine, first, e_uni))
    e_rep = run(rep, "rep", 24)
    assert e_rep[0] == e_uni[-1]
    hot = h.adapt()
    assert hot.flips > calm.flips, (e_uni, e_rep)
    indep = (S.ENGINE_AC_FLAT, S.ENGINE_KEYS)  # the engines whose speed does not depend on the text
    assert e_rep[-1] in indep, e_rep
    settled = min(e_rep.index(e) for e in indep if e in e_rep)
    perf_check(settled <= 12, "text-independent engine reached after %d launches (expected <= 12: one launch in eight reports once an engine has settled): %r" % (settled, e_rep))
    perf_check(hot.ms_per_gib[e_rep[-1]] * 2.0 < max(hot.ms_per_gib[S.ALGO_AC], hot.ms_per_gib[S.ALGO_WM]),
               "settled engine %.3f ms/GiB, text-dependent engines %.3f / %.3f" % (hot.ms_per_gib[e_rep[-1]], hot.ms_per_gib[S.ALGO_AC], hot.ms_per_gib[S.ALGO_WM]))
    # forced engines agree on the counts (and are not overridden)
    forced = 0
    for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT, S.ENGINE_KEYS):
        try:
            h.set_scan_engine(eng)
        except S.SmhError:  # an exact hybrid plan keeps no filter engine; a set served by ONE plain stride-1 image keeps no key table
            assert (eng == S.ALGO_WM and entry == "ac") or eng == S.ENGINE_KEYS
            continue
        forced += 1
        assert _scan(h, rep, n) == counts["rep"] and _scan(h, uni, n) == counts["uni"]
        assert h.adapt().adaptive == 0
    assert forced >= 2
    h.set_scan_engine(-1)
    # full-text counts against the oracle, once
    assert counts["uni"] == O.oracle_ac(pat, m, p, sigma, host["uni"])[0]
    assert counts["rep"] == O.oracle_ac(pat, m, p, sigma, host["rep"])[0]
    h.close()


@pytest.mark.gpu
def test_verify_mode_follows_measured_survivors():
    """8000 patterns of 16 symbols: on uniform text ~1.4 survivors per 4 KiB (verified in registers); on the planted text
    -- one of the patterns recurs in every 64-byte cell -- over 60: the launcher is handed the measured rate and takes the
    staged verify.  Same counts either way, and SMH_WM_TUNE can still force each mode.  (The filter kernels are forced: left to
    itself the handle leaves them on the planted text for its text-independent engines -- the key table since round 5.)"""
    n, m, p, sigma = 64 << 20, 16, 8000, 4
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, S.CORPUS_PLANTED)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    wm.set_scan_engine(S.ALGO_WM)
    uni, pla = _dev_text(n, sigma, S.CORPUS_UNIFORM), _dev_text(n, sigma, S.CORPUS_PLANTED)
    want_uni = O.oracle_ac(pat, m, p, sigma, uni[:n].cpu().numpy())[0]
    want_pla = O.oracle_ac(pat, m, p, sigma, pla[:n].cpu().numpy())[0]
    assert want_pla > n // 64
    for _ in range(3):
        assert _scan(wm, uni, n) == want_uni
    low = wm.adapt()
    assert low.reports >= 2 and low.events_per_4k[S.ALGO_WM] < 8.0
    for _ in range(12):
        assert _scan(wm, pla, n) == want_pla
    high = wm.adapt()
    assert high.events_per_4k[S.ALGO_WM] > 60.0 and high.verify_density * 4096 > 8.0  # the launcher now plans for the staged verify
    wm.close()


@pytest.mark.gpu
def test_small_launches_and_disabled_adaptation(monkeypatch):
    n, m, p, sigma = 4 << 20, 16, 500, 4
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    t = _dev_text(n, sigma, S.CORPUS_UNIFORM)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    want = O.oracle_ac(pat, m, p, sigma, t[:n].cpu().numpy())[0]
    for _ in range(4):
        assert _scan(ac, t, n) == want
    assert ac.adapt().reports == 0  # below 32 MiB a launch is mostly staging and tail: it reports nothing
    ac.close()


PART_SETS = [(4, 1000, 32, S.CORPUS_DNA_REPEATS), (4, 3000, 16, S.CORPUS_UNIFORM), (20, 1000, 8, S.CORPUS_SKEWED), (4, 8000, 16, S.CORPUS_DNA_REPEATS)]


@pytest.mark.parametrize("sigma,p,m,kind", PART_SETS)
def test_text_independent_engine_is_kept_in_parts(sigma, p, m, kind):
    """A set whose automaton does not fit LDS whole keeps the text-independent engine as SEVERAL exact stride-1 automata
    (ac_host.c, end of the compile); a set that would need more than 16 keeps none, and neither does an exact plain plan."""
    pat = S.corpus_patterns(m, p, 7, sigma, 42, 1 << 24, 2, kind)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    info = ac.info()
    assert 2 <= info.flat_parts <= 16 and info.adaptive == 1
    assert ac.adapt().est_ms_per_gib[S.ENGINE_AC_FLAT] > 0.2 * info.flat_parts
    ac.set_scan_engine(S.ENGINE_AC_FLAT)
    assert ac.info().scan_engine == S.ENGINE_AC_FLAT
    ac.close()
    if sigma == 4:  # the Wu-Manber entry point holds the same engine through its automaton handle
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        assert wm.info().adaptive == 1 and wm.adapt().est_ms_per_gib[S.ENGINE_AC_FLAT] > 0.4
        wm.close()
    big = S.corpus_patterns(24, 20000, 7, 4, 42, 1 << 24, 2)
    h = S.AcAutomaton.from_patterns(big, 24, 20000, 4)
    assert h.info().flat_parts == 0
    h.close()


@pytest.mark.parametrize("sigma,p,m,kind", PART_SETS)
def test_parts_partition_the_set_emulated(sigma, p, m, kind):
    """every part scanned by the CPU lane emulator (the kernels' own lane code), summed: the oracle's count of the whole set --
    no pattern lost or entered twice by the cut into runs"""
    import emu_lib as E
    n = 300_000
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, kind)
    text = S.corpus_text(n, 42, sigma, 0, kind)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    got, parts = E.ac_scan_flat_parts(ac, text, 2)
    assert parts == ac.info().flat_parts >= 2
    assert got == O.oracle_ac(pat, m, p, sigma, text)[0] > 0
    ac.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,p,m,kind", PART_SETS)
def test_parts_count_and_positions_equal_the_oracle(sigma, p, m, kind):
    import torch
    n = 12 << 20
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, kind)
    t = _dev_text(n, sigma, kind)
    host = t[:n].cpu().numpy()
    want = O.oracle_ac(pat, m, p, sigma, host)[0]
    assert want > 1000
    handles = [S.AcAutomaton.from_patterns(pat, m, p, sigma)]
    if sigma == 4:
        handles.append(S.WmTables.from_patterns(pat, m, p, sigma))
    for h in handles:
        assert _scan(h, t, n) == want  # the compile's engine
        h.set_scan_engine(S.ENGINE_AC_FLAT)
        assert _scan(h, t, n) == want
        assert _scan(h, t, n - 12345) == O.oracle_ac(pat, m, p, sigma, host[:n - 12345])[0]
        cap = want + 8
        out = torch.zeros(cap, dtype=torch.int64, device="cuda")
        cur = torch.zeros(1, dtype=torch.int64, device="cuda")
        h.positions_device(t.data_ptr(), n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == want
        got = np.sort(out[:want].cpu().numpy())
        h.set_scan_engine(S.ALGO_WM)  # the filter kernels' positions: the same END columns
        out2 = torch.zeros(cap, dtype=torch.int64, device="cuda")
        cur.zero_()
        h.positions_device(t.data_ptr(), n, out2.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == want and np.array_equal(got, np.sort(out2[:want].cpu().numpy()))
        assert len(np.unique(got)) == want  # distinct patterns of one length never share an END column
        h.close()


@pytest.mark.gpu
@pytest.mark.perf
@pytest.mark.parametrize("entry", ["ac", "wm"])
def test_first_look_at_a_long_text(entry):
    """The first tuned count launch of a handle on a device, when the text is 1 GiB or more, scans the first 256 MiB with the
    compile's choice and WAITS for its report (smh_runtime.hip adapt_first_look): on the planted text -- 20-50 x the estimate --
    the other engines scan the same piece into a scratch count and the rest of the text goes to the best of them; on uniform text
    the choice stands.  The count is the same whichever way the launch was cut, and equals every forced engine's."""
    n, m, p, sigma = 1 << 30, 32, 1000, 4
    make = S.AcAutomaton if entry == "ac" else S.WmTables
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, S.CORPUS_PLANTED)
    pla = _dev_text(n, sigma, S.CORPUS_PLANTED)
    h = make.from_patterns(pat, m, p, sigma)
    assert h.info().adaptive == 1
    first = h.info().scan_engine
    c1 = _scan(h, pla, n)
    ad = h.adapt()
    assert ad.engine in (S.ENGINE_AC_FLAT, S.ENGINE_KEYS) and ad.flips == 1 and ad.reports >= 3, (ad.engine, ad.flips, ad.reports)
    perf_check(ad.ms_per_gib[first] > 3.0 * ad.est_ms_per_gib[first], "the compile's engine on the hostile text: %.3f ms/GiB measured, %.3f estimated (expected > 3x)" % (ad.ms_per_gib[first], ad.est_ms_per_gib[first]))
    assert _scan(h, pla, n) == c1  # now one launch, the text-independent engine
    counts = []
    for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT, S.ENGINE_KEYS):
        g = make.from_patterns(pat, m, p, sigma)
        g.set_scan_engine(eng)
        counts.append(_scan(g, pla, n))
        assert g.adapt().flips == 0
        g.close()
    assert counts == [c1] * 4
    # the same text cut at another place: END columns are counted once
    assert _scan(h, pla, n - (300 << 20)) + 0 == _scan(h, pla, n - (300 << 20))
    h.close()
    del pla
    uni = _dev_text(n, sigma, S.CORPUS_UNIFORM)
    u = make.from_patterns(pat, m, p, sigma)
    cu = _scan(u, uni, n)
    au = u.adapt()
    assert au.reports >= 1
    perf_check(au.flips == 0 and au.engine == first, "uniform text after the first look: %d flips, engine %d (compiled: %d)" % (au.flips, au.engine, first))
    u.set_scan_engine(first)
    assert _scan(u, uni, n) == cu
    u.close()
