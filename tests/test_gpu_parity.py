"""Parity tests proper: the real gfx950 kernels, called through the C ABI of libsmatcher_hip.so,
against (i) the reference's golden counts, (ii) the oracle on fresh inputs, and (iii) at BASELINE
sizes, size-independent properties (AC == WM, shard sums == whole, table-walk == tuned kernels).
Integer work: every comparison is bit-exact."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

import cases
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402

pytestmark = pytest.mark.gpu

with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
    VECTORS = json.load(f)
BY_NAME = {v["name"]: v for v in VECTORS}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if S.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the MI355X box")


@pytest.mark.parametrize("vec", VECTORS, ids=[v["name"] for v in VECTORS])
def test_kernels_match_reference_vectors(vec):
    text, pat = cases.build(vec)
    p, m, sigma, want = vec["p"], vec["m"], vec["sigma"], vec["count_ac"]
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == want
    assert ac.count_host(text, S.VARIANT_TABLE)[0] == want
    assert wm.count_host(text, S.VARIANT_TUNED)[0] == want == vec["count_wu2"]
    assert wm.count_host(text, S.VARIANT_TABLE)[0] == want
    if wm.info().scan_engine == S.ALGO_AC:
        # a small-alphabet set of long patterns: the handle chose the automaton kernels; its own too
        wm.set_scan_engine(S.ALGO_WM)
        assert wm.info().scan_engine == S.ALGO_WM and wm.count_host(text, S.VARIANT_TUNED)[0] == want


@pytest.mark.parametrize("seed", range(10))
def test_kernels_match_oracle_on_fresh_inputs(seed):
    rng = np.random.RandomState(1234 + seed)
    sigma = [2, 4, 8, 20, 128, 256][seed % 6]
    m = int(rng.randint(3, 66))
    p = int(rng.randint(1, 3000))
    n = int(rng.randint(m, 3_000_000))
    text = O.gen_text(n, 500 + seed, sigma)
    pat = O.gen_patterns_mixed(m, p, 600 + seed, sigma, 500 + seed, n, 2)
    want_ac, _ = O.oracle_ac(pat, m, p, sigma, text)
    want_wm, _ = O.oracle_wu(pat, m, p, sigma, text)
    assert want_ac == want_wm
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    for variant in (S.VARIANT_TUNED, S.VARIANT_TABLE):
        assert ac.count_host(text, variant)[0] == want_ac
        assert wm.count_host(text, variant)[0] == want_wm


def test_legacy_api_reads_like_the_reference_driver(capfd):
    """preproc_* / search_* / cuda_* exactly as main.c:125-157, 268-298, 582-648 calls them."""
    vec = BY_NAME["dense_dna"]
    text, pat = cases.build(vec)
    n, p, m, sigma, want = vec["n"], vec["p"], vec["m"], vec["sigma"], vec["count_ac"]
    tp = text.ctypes.data_as(S.u8p)
    # --- multiac
    t = O.ACTables(m, p, sigma)
    rows = np.zeros((p, m + 1), dtype=np.uint8)
    rows[:, :m] = pat.reshape(p, m)
    arr = (S.u8p * p)()
    for j in range(p):
        arr[j] = C.cast(rows[j].ctypes.data, S.u8p)
    tab = S.lib.preproc_ac(arr, m, p, sigma, t.state_transition.ctypes.data_as(S.i32p),
                           t.state_supply.ctypes.data_as(S.u32p), t.state_final.ctypes.data_as(S.u32p))
    assert S.lib.search_ac(tp, n, tab) == want
    S.lib.free_ac(tab, sigma)
    capfd.readouterr()
    S.lib.smh_host_path_release()
    builds = S.lib.smh_legacy_handle_builds()
    for k in range(1, 6):
        getattr(S.lib, "cuda_ac%d" % k)(m, tp, n, p, sigma, t.state_transition.ctypes.data_as(S.i32p),
                                        t.state_supply.ctypes.data_as(S.u32p), t.state_final.ctypes.data_as(S.u32p))
    # round 6: main.c:583-592's five back-to-back calls on the same tables compile ONE handle and send ONE table set
    assert S.lib.smh_legacy_handle_builds() == builds + 1
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("Kernel")]
    assert len(lines) == 5
    for k, ln in enumerate(lines, 1):
        parts = ln.split("\t")  # "Kernel K matches \t%i\t time \t%f"  (cuda/cuda_ac.cu:675)
        assert parts[0] == "Kernel %d matches " % k and int(parts[1]) == want and float(parts[3]) > 0
    # --- multiwm2 / multiwm
    S.lib.wu_determine_shiftsize(sigma)
    w = O.WMTables(m, p, sigma, S.shiftsize_global())
    S.lib.preproc_wu2(pat.ctypes.data_as(S.u8p), m, p, sigma, 3, *w.ptrs())
    assert S.lib.search_wu2(pat.ctypes.data_as(S.u8p), m, p, tp, n, *w.ptrs()) == want
    prow = np.ascontiguousarray(pat.reshape(p, m))
    parr = (S.u8p * p)()
    for j in range(p):
        parr[j] = C.cast(prow[j].ctypes.data, S.u8p)
    assert S.lib.search_wu(parr, m, p, tp, n, *w.ptrs()) == want
    builds = S.lib.smh_legacy_handle_builds()
    for k in range(1, 6):
        secs = C.c_double(0)
        got = getattr(S.lib, "cuda_wm%d" % k)(pat.ctypes.data_as(S.u8p), m, tp, n, p, sigma, 3, *w.ptrs(), C.byref(secs))
        assert got == want and secs.value > 0
    # main.c:623-648: ONE handle serves all five (search_wu above came with another pattern pointer: cuda_wm1 compiled, 2..5 reused)
    assert S.lib.smh_legacy_handle_builds() == builds + 1
    # the caller rewrites its arrays IN PLACE (same pointers): other contents, another handle, the other count
    half = p // 2
    pat2 = pat.copy()
    pat2[half * m:] = pat2[:(p - half) * m]  # the second half repeats the first: fewer distinct patterns
    pat[:] = pat2
    w2 = O.WMTables(m, p, sigma, S.shiftsize_global())
    S.lib.preproc_wu2(pat.ctypes.data_as(S.u8p), m, p, sigma, 3, *w2.ptrs())
    for name in ("SHIFT", "PREFIX_value", "PREFIX_index", "PREFIX_size"):
        getattr(w, name)[:] = getattr(w2, name)
    want2 = O.count_bruteforce(pat, m, p, text)
    secs = C.c_double(0)
    assert S.lib.cuda_wm5(pat.ctypes.data_as(S.u8p), m, tp, n, p, sigma, 3, *w.ptrs(), C.byref(secs)) == want2 != want
    assert S.lib.smh_legacy_handle_builds() == builds + 2
    S.lib.smh_host_path_release()


def test_device_resident_text_and_streams():
    import torch
    dev = torch.device("cuda", 0)
    n, sigma, m, p = (1 << 24) + 37, 4, 8, 1000
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(stream)) == 0
    host = S.corpus_text(n, 42, sigma)
    assert np.array_equal(text[:n].cpu().numpy(), host)  # device generator == host generator == oracle generator
    assert np.array_equal(host, O.gen_text(n, 42, sigma))
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    want, _ = O.oracle_ac(pat, m, p, sigma, host)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, stream)
    ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, stream)        # counts accumulate
    wm.scan_device(text.data_ptr(), n, cnt.data_ptr() + 8, S.VARIANT_TUNED, stream)
    with torch.cuda.stream(side):
        ac.scan_device(text.data_ptr(), n, cnt.data_ptr() + 16, S.VARIANT_TABLE, side.cuda_stream)
        wm.scan_device(text.data_ptr(), n, cnt.data_ptr() + 24, S.VARIANT_TABLE, side.cuda_stream)
    torch.cuda.synchronize()
    assert cnt.tolist() == [2 * want, want, want, want]
    # unaligned device pointer is refused, not mis-scanned
    with pytest.raises(S.SmhError):
        ac.scan_device(text.data_ptr() + 4, n - 4, cnt.data_ptr(), S.VARIANT_TUNED, stream)


def test_stream_read_probe_variants_read_every_byte():
    """the five read-only probes (bench.py `stream_read`) XOR all dwords of the buffer: same value, the host's"""
    import torch
    dev = torch.device("cuda", 0)
    n = (96 << 20) + 4096 * 7  # a multiple of the 4 KiB wave-chunk, not of the two-chunk stride
    buf = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
    want = int(np.bitwise_xor.reduce(buf.cpu().numpy().view(np.uint32)))
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for v in range(5):
        out.zero_()
        assert S.lib.smh_stream_read_probe_variant(C.c_void_p(buf.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(stream), v) == 0
        torch.cuda.synchronize()
        assert int(out.item()) == want, v
    assert S.lib.smh_stream_read_probe_variant(C.c_void_p(buf.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(stream), 5) != 0


@pytest.mark.parametrize("name", ["big_dfa", "ascii_5_20", "mx_s256_m32_p1000"])
def test_dfa_larger_than_lds(name):
    """Automata that do not fit LDS are cut at depth K; candidates are verified in HBM -- same counts."""
    vec = BY_NAME[name]
    text, pat = cases.build(vec)
    ac = S.AcAutomaton.from_patterns(pat, vec["m"], vec["p"], vec["sigma"])
    info = ac.info()
    assert info.lds_rows < info.rows and not info.scan_exact and info.scan_depth < vec["m"]
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == vec["count_ac"]  # whichever engine the handle chose
    ac.set_scan_plan(info.scan_stride, info.scan_depth)               # the automaton kernels, same plan
    assert ac.info().scan_engine == S.ALGO_AC
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == vec["count_ac"]


@pytest.mark.parametrize("name,engine", [("ascii_5_20", 1), ("ascii_m5", 1), ("mx_s256_m16_p1000", 1), ("dense_dna", 0),
                                         ("big_dfa", 1), ("kat_1m_100x8", 0)])
def test_scan_engine_choice(name, engine):
    """Sets whose best LDS automaton would be verify-bound (alphabet 256: K = 1) -- and, round 3, depth-cut plans the
    pair-gram filter is estimated to beat (big_dfa) -- are scanned by the
    suffix-filter kernels behind the same AC entry points -- counts and positions identical; forcing a
    plan switches back to the automaton kernels, (0, 0) restores the choice."""
    import torch
    vec = BY_NAME[name]
    text, pat = cases.build(vec)
    m, p, sigma, want = vec["m"], vec["p"], vec["sigma"], vec["count_ac"]
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    assert ac.info().scan_engine == (S.ALGO_WM if engine else S.ALGO_AC)
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == want
    dev = torch.device("cuda", 0)
    d_text = torch.zeros(len(text) + 64, dtype=torch.uint8, device=dev)
    d_text[:len(text)] = torch.from_numpy(text).to(dev)
    wantpos = O.positions_bruteforce(pat, m, p, text)
    for forced in (False, True):
        if forced:
            ac.set_scan_plan(1, ac.info().scan_depth)
            assert ac.info().scan_engine == S.ALGO_AC and ac.count_host(text, S.VARIANT_TUNED)[0] == want
        out = torch.zeros(want + 4, dtype=torch.int64, device=dev)
        cur = torch.zeros(1, dtype=torch.int64, device=dev)
        ac.positions_device(d_text.data_ptr(), len(text), out.data_ptr(), want + 4, cur.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert int(cur.item()) == want and np.array_equal(np.sort(out[:want].cpu().numpy()), wantpos)
    ac.set_scan_plan(0, 0)
    assert ac.info().scan_engine == (S.ALGO_WM if engine else S.ALGO_AC)
    # a handle built from the reference tables reads the patterns back from the goto trie
    t = O.oracle_ac(pat, m, p, sigma)[1]
    ac2 = S.AcAutomaton.from_tables(t.state_transition, t.state_supply, t.state_final, m * p + 1, sigma, m)
    assert ac2.info().scan_engine == ac.info().scan_engine and ac2.count_host(text, S.VARIANT_TUNED)[0] == want


@pytest.mark.parametrize("name", ["dense_dna", "kat_1m_100x8", "mx_s4_m16_p1000", "mx_s4_m32_p100", "edge_m65",
                                  "overlap_zeros", "mx_s20_m8_p1000", "big_dfa", "mx_s256_m8_p1000", "ascii_m5",
                                  "mx_s128_m16_p100"])
def test_every_scan_plan_gives_the_same_count(name):
    """Stride 1 / 2 / hybrid, exact (K = m) and depth-cut (K < m) automata, down to K = 1 where nearly
    every position is a candidate and the per-wave queue overflows and drains constantly."""
    vec = BY_NAME[name]
    text, pat = cases.build(vec)
    m, sigma = vec["m"], vec["sigma"]
    ac = S.AcAutomaton.from_patterns(pat, m, vec["p"], sigma)
    tried = 0
    for stride in ((1, 2) if sigma == 4 else (1,)):
        for depth in sorted({1, 2, 3, max(1, m // 2), max(1, m - 1), min(m, 65)}):
            if depth > min(m, 65):
                continue
            try:
                ac.set_scan_plan(stride, depth)
            except S.SmhError:
                continue
            tried += 1
            assert ac.count_host(text, S.VARIANT_TUNED)[0] == vec["count_ac"], (stride, depth)
    if sigma == 4:  # hybrid stride-2 images: (K, D) = scan depth, depth of the full rows
        for K in sorted({min(m, 65), max(4, m - 1), max(4, m // 2), 5}):
            for D in sorted({1, 2, max(1, K - 3)}):
                if K > min(m, 65) or D > K - 3:
                    continue
                try:
                    ac.set_scan_plan(3, K | (D << 8))
                except S.SmhError:
                    continue
                assert ac.info().scan_full_rows > 0
                assert ac.count_host(text, S.VARIANT_TUNED)[0] == vec["count_ac"], ("hybrid", K, D)
    assert tried >= (3 if sigma <= 20 else 1)  # alphabet 128 / 256: only K = 1 fits LDS


@pytest.mark.parametrize("tune", ["clamp=0", "clamp=1", "clamp=0,nch=3", "clamp=1,nch=3", "clamp=0,nch=2", "clamp=1,nch=2", "nch=1"])
def test_hybrid_kernel_instantiations_agree(tune, knob):
    """Every instantiation of the hybrid-image kernels gives the reference's count: the unclamped full-row lookup (run only
    after the per-device probe smh_lds_oob_reads_zero has seen out-of-range LDS reads return 0) and its clamped twin, two
    chains with the register prefetch and three without, exact and depth-cut plans, 4 MiB of text with planted matches."""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    knob.ac(tune)
    sigma, n = 4, (4 << 20) + 12345
    text = O.gen_text(n, 77, sigma)
    for m, p, plans in ((12, 1000, [(3, 12 | (9 << 8)), (3, 12 | (6 << 8)), (3, 9 | (5 << 8))]),
                        (16, 1000, [(0, 0), (3, 12 | (9 << 8)), (3, 16 | (8 << 8)), (3, 13 | (7 << 8))]),
                        (32, 600, [(0, 0), (3, 12 | (8 << 8)), (3, 17 | (9 << 8))])):
        pat = O.gen_patterns_mixed(m, p, 300 + m, sigma, 77, n, 2)
        want, _ = O.oracle_ac(pat, m, p, sigma, text)
        assert want > p // 4
        ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
        ran = 0
        for stride, depth in plans:
            try:
                ac.set_scan_plan(stride, depth)
            except S.SmhError:
                continue
            if stride == 3:
                assert ac.info().scan_full_rows > 0
            assert ac.count_host(text, S.VARIANT_TUNED)[0] == want, (tune, m, stride, depth)
            ran += 1
        assert ran >= 2


@pytest.mark.parametrize("m,p", [(8, 8000), (8, 300), (5, 700), (3, 50)])
def test_dense_plan_on_the_gpu(m, p):
    """The dense plan of the automaton engine (smh_ac_info.scan_dense): counts and positions against the oracle, chosen
    (8000 8-mers) and forced; the ordinary plans of the same handle must agree."""
    import torch
    sigma, n = 4, (6 << 20) + 4321
    text = O.gen_text(n, 91, sigma)
    pat = O.gen_patterns_mixed(m, p, 400 + m, sigma, 91, n, 2)
    want, _ = O.oracle_ac(pat, m, p, sigma, text)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    if (m, p) == (8, 8000):
        assert ac.info().scan_dense == 1 and ac.info().scan_engine == S.ALGO_AC
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == want
    ac.set_scan_plan(4, 0)
    assert ac.info().scan_dense == 1
    assert ac.count_host(text, S.VARIANT_TUNED)[0] == want
    dev = torch.device("cuda", 0)
    dtext = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    dtext[:n] = torch.from_numpy(text).to(dev)
    cap = want + 16
    out = torch.zeros(cap, dtype=torch.int64, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    ac.positions_device(dtext.data_ptr(), n, out.data_ptr(), cap, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == want
    ref = O.positions_bruteforce(pat, m, p, text)  # END columns
    assert len(ref) == want and np.array_equal(np.sort(out[:want].cpu().numpy()), np.sort(ref))
    ac.set_scan_plan(1, 0)
    assert ac.info().scan_dense == 0 and ac.count_host(text, S.VARIANT_TUNED)[0] == want


def test_baseline_size_properties():
    """BASELINE configs[1]/[2] size: 1 GiB DNA text in HBM.  The oracle cannot scan that in seconds,
    so: (a) oracle on a 32 MiB slice, (b) AC == WM on the full text, (c) sum over the 8 byte-range
    shards (main.c:467-477) == whole, (d) table-walking kernels == tuned kernels on a 128 MiB slice."""
    import torch
    dev = torch.device("cuda", 0)
    n, sigma = 1 << 30, 4
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(stream)) == 0
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def run(obj, off, length, variant=S.VARIANT_TUNED):
        cnt.zero_()
        obj.scan_device(text.data_ptr() + off, length, cnt.data_ptr(), variant, stream)
        torch.cuda.synchronize()
        return int(cnt.item())

    for m, p in ((8, 1000), (16, 1000), (32, 1000), (8, 10000)):
        pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
        ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
        wm = S.WmTables.from_patterns(pat, m, p, sigma)
        slice_n = 32 << 20
        host = S.corpus_text(slice_n, 42, sigma)
        want, _ = O.oracle_ac(pat, m, p, sigma, host)
        assert run(ac, 0, slice_n) == want and run(wm, 0, slice_n) == want
        whole = run(ac, 0, n)
        assert whole == run(wm, 0, n)
        assert whole >= p // 2  # every pattern sampled from the text occurs at least once
        parts = 0
        for i in range(8):
            b, e = S.shard_range(n, 8, i, m)
            parts += run(ac if i % 2 == 0 else wm, b, e - b)
        assert parts == whole
        t_n = 128 << 20
        tuned = run(ac, 0, t_n)
        assert run(ac, 0, t_n, S.VARIANT_TABLE) == tuned and run(wm, 0, t_n, S.VARIANT_TABLE) == tuned


def test_text_longer_than_4gib():
    """64-bit lengths (the reference's int n stops at 2 GiB, smatcher.h:90): a 4 GiB + 12345 byte text,
    whole == two halves that overlap by m-1."""
    import torch
    dev = torch.device("cuda", 0)
    n, sigma, m, p = (1 << 32) + 12345, 4, 8, 1000
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(stream)) == 0
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)

    def run(obj, off, length):
        cnt.zero_()
        obj.scan_device(text.data_ptr() + off, length, cnt.data_ptr(), S.VARIANT_TUNED, stream)
        torch.cuda.synchronize()
        return int(cnt.item())

    whole = run(ac, 0, n)
    assert whole == run(wm, 0, n)
    b0, e0 = S.shard_range(n, 2, 0, m)
    b1, e1 = S.shard_range(n, 2, 1, m)
    b1a = b1 - (b1 % 16)  # keep the device pointer 16-byte aligned: start a little earlier ...
    extra = run(ac, b1a, b1 - b1a + m - 1) if b1a != b1 else 0  # ... and subtract what the overlap adds
    assert run(ac, b0, e0 - b0) + run(wm, b1a, e1 - b1a) - extra == whole
    # the tail of the text (beyond 2^32) really is scanned: plant-free check against the oracle
    tail_off = (1 << 32) - 4096
    host_tail = S.corpus_text(n - tail_off, 42, sigma, offset=tail_off)
    want_tail, _ = O.oracle_ac(pat, m, p, sigma, host_tail)
    assert run(ac, tail_off, n - tail_off) == want_tail


@pytest.mark.parametrize("m", [5, 12, 20])
def test_ascii_100k_patterns_against_bruteforce(m):
    """BASELINE configs[4] shape (256-symbol alphabet, 100 000 patterns, lengths 5-20) at a size the
    definition-level brute force finishes in seconds: WM hashed filter + survivor queue + verify, WM table
    walk, and AC with the automaton cut at a shallow depth (alphabet 256 rows are 512 B: K = 1)."""
    sigma, p, n = 256, 100000, 6 * 1024 * 1024 + 77
    text = S.corpus_text(n, 42, sigma)
    pat = S.corpus_patterns(m, p, 9, sigma, 42, n, 2)
    want = O.count_bruteforce(pat, m, p, text)
    assert want > p // 3  # about half the patterns are cut from the text
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    info = wm.info()
    assert info.filter_hashed and not info.filter_exact and info.distinct <= p
    assert wm.count_host(text, S.VARIANT_TUNED)[0] == want
    assert wm.count_host(text[:1 << 20], S.VARIANT_TABLE)[0] == O.count_bruteforce(pat, m, p, text[:1 << 20])
    if m <= 12:
        # the automaton over the first 30 000 patterns (a 0.3-3 GB host compile stays in seconds)
        pa = 30000
        ac = S.AcAutomaton.from_patterns(pat[:pa * m], m, pa, sigma)
        assert not ac.info().scan_exact
        assert ac.count_host(text[:1 << 20], S.VARIANT_TUNED)[0] == O.count_bruteforce(pat[:pa * m], m, pa, text[:1 << 20])


@pytest.mark.parametrize("lane0", [0, 1])
@pytest.mark.parametrize("m,p", [(11, 3000), (13, 500), (16, 3000), (17, 8000), (18, 3000), (21, 200), (27, 5000), (32, 8000), (33, 3000)])
def test_pair_gram_lane0_state_on_the_gpu(m, p, lane0, knob):
    """Pair-gram filter: the shift-or state that lane 0 of a wave-chunk inherits is either assumed (every plane alive)
    or worked out from the 16 / 32 bytes in front of the chunk (wm_lane.h; the launcher's choice, forced both ways
    here).  Occurrences END in each of the first 15 columns of a chunk -- the columns that state decides -- in every
    chunk of the text, with enough patterns that lane 0 has early flags in most chunks."""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    knob.wm("gram=1,lane0=%d" % lane0)
    rng = np.random.RandomState(77 * m + p)
    n = 48 * 4096 + 123
    text = rng.randint(0, 4, size=n).astype(np.uint8)
    pat = rng.randint(0, 4, size=(p, m)).astype(np.uint8)
    for k in range(1, 48):
        end = k * 4096 + (k % 15)
        text[end - m + 1:end + 1] = pat[(7 * k) % p]
    wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, 4)
    assert wm.info().gram_planes == min(15, m - 6)
    if wm.info().scan_engine != S.ALGO_WM:
        wm.set_scan_engine(S.ALGO_WM)
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    assert want >= 47
    assert wm.count_host(text, S.VARIANT_TUNED)[0] == want
    import torch
    d_text = torch.from_numpy(text).cuda()
    out = torch.zeros(want + 8, dtype=torch.int64, device="cuda")
    cur = torch.zeros(1, dtype=torch.int64, device="cuda")
    wm.positions_device(d_text.data_ptr(), n, out.data_ptr(), want + 8, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == want
    assert sorted(out[:want].tolist()) == O.positions_bruteforce(pat.reshape(-1), m, p, text).tolist()


@pytest.mark.parametrize("kind,m,p", [(6, 5, 3000), (6, 6, 100000), (6, 8, 30000), (6, 17, 1000), (6, 33, 2000), (2, 5, 3000), (2, 12, 30000),
                                      (2, 17, 100000), (2, 18, 1000), (2, 33, 2000)])
def test_byte_gram_forms_with_the_staged_verify_on_the_gpu(kind, m, p, knob):
    """Round 4: the byte forms take their survivors' windows from L2 (a two-stage pipeline across chunks, wm_lane.h
    smh_wm_l2_columns; what the default / hbm_windows / staged / in_registers ids below run for kinds 2 and 6); this is the
    round-3 staged verify (the chunk copied to LDS) on the same texts."""
    test_gram_filter_forms_on_the_gpu(kind, 256, m, p, ",l2=0", knob)


@pytest.mark.parametrize("form", [6, 9])
@pytest.mark.parametrize("m,p", [(5, 3000), (6, 100000), (7, 100)])
def test_flat_byte_grams_two_bits_per_gram_on_the_gpu(m, p, form, knob):
    """the flat form of 5..7-byte patterns with two bits per gram forced (round 4; form 9, round 6: in the 143.9 KiB table): same
    text, same checks as below"""
    test_gram_filter_forms_on_the_gpu(form, 256, m, p, ",flatk=2", knob)


@pytest.mark.parametrize("stage", ["", ",stage=0", ",regv=0", ",regv=1", ",l2=1"], ids=["default", "hbm_windows", "staged", "in_registers", "windows_from_l2"])
@pytest.mark.parametrize("kind,sigma,m,p", [(1, 4, 11, 40), (1, 4, 16, 300), (1, 4, 17, 6000), (1, 4, 33, 50), (1, 4, 40, 50),
                                            (3, 4, 11, 200), (3, 4, 16, 20000), (3, 4, 32, 500), (2, 256, 5, 3000),
                                            (6, 256, 5, 3000), (6, 256, 6, 100000), (6, 256, 7, 100), (6, 256, 8, 30000), (6, 256, 17, 1000), (6, 256, 33, 2000),
                                            (5, 4, 11, 30), (5, 4, 16, 8000), (5, 4, 17, 6000), (5, 4, 18, 300), (5, 4, 23, 20000), (5, 4, 33, 50),
                                            (2, 256, 12, 30000), (2, 256, 17, 100000), (2, 256, 18, 1000), (2, 256, 33, 2000),
                                            (2, 256, 34, 2000), (2, 128, 7, 100),
                                            (8, 256, 5, 3000), (8, 256, 12, 30000), (8, 256, 17, 100000), (8, 256, 20, 100000), (8, 256, 33, 2000), (8, 20, 10, 500),
                                            (9, 256, 5, 3000), (9, 256, 6, 100000), (9, 256, 8, 30000), (9, 256, 9, 100000), (9, 256, 17, 1000), (9, 256, 33, 2000),
                                            (11, 20, 8, 10000), (11, 20, 6, 300), (11, 20, 12, 30000), (11, 256, 7, 3000), (11, 256, 17, 100000), (11, 128, 33, 2000)])
def test_gram_filter_forms_on_the_gpu(kind, sigma, m, p, stage, knob):
    """Each q-gram shift-or form forced (development knob), with the staged verify (window hashes from the LDS copy
    of the chunk, 16- and 32-byte halo, m = 17 / 33 at their limits, m = 34 / 40 beyond them), with the pair form's
    in-register verify forced on and off (round 3; the stretch where every column survives gives a lane 64 rounds of it),
    with windows re-read from HBM and -- late round 6, the 4-letter forms too -- through the windows-from-L2 pipeline; texts with planted occurrences at chunk / segment boundaries and a stretch where EVERY column
    survives the filter (a pattern repeated back to back), so lists overflow and are flushed mid-chunk."""
    S = knob.T  # the testing build: the development knobs exist only there (csrc/smh_tune.h)
    knob.wm("gram=%d%s" % (kind, stage))
    rng = np.random.RandomState(1000 * kind + m)
    n = 9 * 4096 + 777
    text = rng.randint(0, sigma, size=n).astype(np.uint8)
    pat = rng.randint(0, sigma, size=(p, m)).astype(np.uint8)
    pat[1] = pat[0][0]  # a one-symbol run: every column inside a run of it matches
    text[5 * 4096 - 700:5 * 4096 + 900] = pat[0][0]
    rep = np.tile(pat[2], 3000 // m + 2)[:3000]  # pattern 2 back to back: one match every m columns, all grams in planes
    text[7 * 4096 - 1500:7 * 4096 + 1500] = rep
    for i, off in enumerate([0, 300, 640 - m // 2, 4096 - m // 2, 8191, 8192 + 2 * m + 64, 3 * 4096 - 1, 3 * 4096, n - m]):
        text[off:off + m] = pat[(7 * i + 3) % p]
    pat[p // 2] = pat[3]  # a duplicate pattern: a column is counted once
    wm = S.WmTables.from_patterns(pat.reshape(-1), m, p, sigma)
    assert wm.info().gram_planes == min({1: 15, 5: 16}.get(kind, 8), m - {1: 6, 3: 7, 2: 2, 5: 7, 6: 2, 8: 2, 9: 2, 11: 3}[kind]) and wm.info().gram_kind == kind
    if wm.info().scan_engine != S.ALGO_WM:
        wm.set_scan_engine(S.ALGO_WM)
    want = O.count_bruteforce(pat.reshape(-1), m, p, text)
    assert want >= 1600
    assert wm.count_host(text, S.VARIANT_TUNED)[0] == want
    import torch
    d_text = torch.from_numpy(text).cuda()
    out = torch.zeros(want + 8, dtype=torch.int64, device="cuda")
    cur = torch.zeros(1, dtype=torch.int64, device="cuda")
    wm.positions_device(d_text.data_ptr(), n, out.data_ptr(), want + 8, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == want
    assert sorted(out[:want].tolist()) == O.positions_bruteforce(pat.reshape(-1), m, p, text).tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["wm", "ac"])
def test_short_texts_of_a_big_table_handle_go_through_the_key_image(entry):
    """Late round 6: a handle that keeps a key image beside a big-table filter (8000 protein patterns of 8: the four-byte-gram
    form) scans texts under 32 MiB with the image -- a launch over a few MiB is mostly its table staging -- and longer ones with
    the filter; the count is the definition's either way, and the handle's engine is left alone."""
    import torch
    sigma, m, p = 20, 8, 8000
    n_small, n_big = 3 << 20, 40 << 20
    text = S.corpus_text(n_big, 42, sigma)
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n_big, 2)
    h = (S.WmTables if entry == "wm" else S.AcAutomaton).from_patterns(pat, m, p, sigma)
    info = h.info()
    assert info.key_slots > 0 and info.adaptive == 1
    if entry == "wm":
        assert info.gram_kind == 11
    d_text = torch.from_numpy(text).cuda()
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for n in (n_small, n_big, n_small):
        cnt.zero_()
        h.scan_device(d_text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
        torch.cuda.synchronize()
        assert int(cnt.item()) == O.count_bruteforce(pat, m, p, text[:n]), n
    assert h.adapt().engine == (S.ALGO_WM if entry == "wm" else h.adapt().engine)  # the short launches did not move the handle's engine
