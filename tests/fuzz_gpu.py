#!/usr/bin/env python3
"""Differential fuzzing on the GPU (run by tests/test_fuzz_gpu.py, or by hand for more cases): random pattern sets
with long shared prefixes / suffixes, duplicates and text-cut patterns over random alphabets and sizes;
every engine and forced plan must give the oracle's count, positions must equal the brute force.
usage: python tests/fuzz_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (before the library: one HIP runtime per process)
import oracle_lib as O  # noqa: E402  (the checker)
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402


def make_case(rng):
    sigma = int(rng.choice([2, 4, 4, 4, 8, 20, 128, 256]))
    m = int(rng.randint(3, 41)) if rng.rand() < 0.75 else int(rng.randint(3, 9))
    p = int(rng.choice([1, 2, 7, 50, 300, 1000, 3000] if not os.environ.get("FUZZ_BIG") else [3000, 8000, 20000]))
    n = int(rng.randint(m, 1_500_000))
    text = rng.randint(0, sigma, size=n).astype(np.uint8)
    if rng.rand() < 0.2:
        text[:] = text[0]  # constant text: overlapping matches
    pats = np.zeros((p, m), dtype=np.uint8)
    for j in range(p):
        kind = rng.rand()
        if j > 0 and kind < 0.35:      # shares a long prefix with an earlier pattern
            pats[j] = pats[rng.randint(0, j)]
            k = rng.randint(1, max(2, m // 3))
            pats[j, m - k:] = rng.randint(0, sigma, size=k)
        elif j > 0 and kind < 0.5:     # shares a long suffix
            pats[j] = pats[rng.randint(0, j)]
            k = rng.randint(1, max(2, m // 3))
            pats[j, :k] = rng.randint(0, sigma, size=k)
        elif j > 0 and kind < 0.55:    # duplicate
            pats[j] = pats[rng.randint(0, j)]
        elif kind < 0.8 and n > m:     # cut from the text
            off = rng.randint(0, n - m + 1)
            pats[j] = text[off:off + m]
        else:
            pats[j] = rng.randint(0, sigma, size=m)
    return sigma, m, p, text, np.ascontiguousarray(pats.reshape(-1))


def run(cases, seed, verbose=True):
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda", 0)
    checks = 0
    for it in range(cases):
        sigma, m, p, text, pat = make_case(rng)
        want = O.count_bruteforce(pat, m, p, text)
        wantpos = O.positions_bruteforce(pat, m, p, text)
        tag = "case %d sigma=%d m=%d p=%d n=%d want=%d" % (it, sigma, m, p, len(text), want)
        d_text = torch.zeros(len(text) + 64, dtype=torch.uint8, device=dev)
        d_text[:len(text)] = torch.from_numpy(text).to(dev)

        def positions(obj):
            out = torch.zeros(want + 4, dtype=torch.int64, device=dev)
            cur = torch.zeros(1, dtype=torch.int64, device=dev)
            obj.positions_device(d_text.data_ptr(), len(text), out.data_ptr(), want + 4, cur.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert int(cur.item()) == want, (tag, "positions cursor", int(cur.item()))
            assert np.array_equal(np.sort(out[:want].cpu().numpy()), wantpos), (tag, "positions")

        ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
        assert ac.count_host(text)[0] == want, (tag, "ac auto", ac.info().scan_engine)
        assert ac.count_host(text, S.VARIANT_TABLE)[0] == want, (tag, "ac table")
        positions(ac)
        if ac.info().scan_engine == S.ALGO_WM:  # the entry point chose the filter engine: the automaton kernels on the same plan
            ac.set_scan_engine(S.ALGO_AC)
            assert ac.count_host(text)[0] == want, (tag, "ac own kernels", ac.info().scan_stride, ac.info().scan_depth)
            positions(ac)
            ac.set_scan_engine(-1)
            checks += 2
        if ac.info().flat_parts:  # the text-independent engine: the set as that many exact stride-1 automata, one launch each
            ac.set_scan_engine(S.ENGINE_AC_FLAT)
            assert ac.count_host(text)[0] == want, (tag, "ac flat parts", ac.info().flat_parts)
            positions(ac)
            ac.set_scan_engine(-1)
            checks += 2
        for eng, slots in ((S.ENGINE_KEYS, ac.info().key_slots), (S.ENGINE_HASH, ac.info().hash_slots)):  # round 5: the key table / the window-hash engine
            if slots:
                ac.set_scan_engine(eng)
                assert ac.count_host(text)[0] == want, (tag, "ac engine", eng)
                positions(ac)
                ac.set_scan_engine(-1)
                checks += 2
        plans = [(1, min(m, 33)), (1, max(1, m // 2)), (1, 2)]
        if sigma == 4:
            plans += [(2, min(m, 33)), (2, max(1, m // 2)), (3, min(m, 33) | (1 << 8)), (3, min(m, 33) | (3 << 8)),
                      (3, max(4, m // 2) | (2 << 8)), (3, min(m, 20))]
        if sigma == 4 and 3 <= m <= 8:
            plans.append((4, 0))  # the dense plan: the automaton completed to all 4^m strings (smh_ac_info.scan_dense)
        for stride, depth in plans:
            try:
                ac.set_scan_plan(stride, depth)
            except S.SmhError:
                continue
            assert ac.count_host(text)[0] == want, (tag, "ac plan", stride, depth & 255, depth >> 8)
            positions(ac)
            checks += 2
        if sigma in (2, 4, 8, 20, 128, 256):
            wm = S.WmTables.from_patterns(pat, m, p, sigma)
            assert wm.count_host(text)[0] == want, (tag, "wm auto", wm.info().scan_engine)
            assert wm.count_host(text, S.VARIANT_TABLE)[0] == want, (tag, "wm table")
            positions(wm)
            try:  # ... and through the Wu-Manber handle, when it keeps an automaton that brought parts along
                wm.set_scan_engine(S.ENGINE_AC_FLAT)
                assert wm.count_host(text)[0] == want, (tag, "wm flat parts")
                positions(wm)
                wm.set_scan_engine(-1)
                checks += 2
            except S.SmhError:
                pass
            for eng, slots in ((S.ENGINE_KEYS, wm.info().key_slots), (S.ENGINE_HASH, wm.info().hash_slots)):
                if slots:
                    wm.set_scan_engine(eng)
                    assert wm.count_host(text)[0] == want, (tag, "wm engine", eng)
                    positions(wm)
                    wm.set_scan_engine(-1)
                    checks += 2
            if wm.info().scan_engine == S.ALGO_AC:
                wm.set_scan_engine(S.ALGO_WM)
                assert wm.count_host(text)[0] == want, (tag, "wm own kernels")
                positions(wm)
        sh = S.ShTrie.from_patterns(pat, m, p, sigma)
        assert sh.count_host(text)[0] == want and sh.count_host(text[:200000], None, S.VARIANT_TABLE)[0] == \
            O.count_bruteforce(pat, m, p, text[:200000]), (tag, "sh")
        # SBOM rows hold 199 pattern ids per state: skip sets where more than that many patterns can coincide
        flat = pat.reshape(p, m)
        if max(np.unique(flat, axis=0, return_counts=True)[1]) < 199:
            try:
                sb = S.SbomOracle.from_patterns(pat, m, p, sigma)
            except S.SmhError:
                sb = None
            if sb is not None:
                assert sb.count_host(text)[0] == want, (tag, "sbom tuned")
                assert sb.count_host(text[:200000], S.VARIANT_TABLE)[0] == O.count_bruteforce(pat, m, p, text[:200000]), (tag, "sbom table")
        # a mixed-length set over the same text: lengths m .. m+5 (one-pass or per-class, both algorithms)
        if it % 3 == 0 and len(text) > m + 8:
            nl = int(rng.randint(2, 6))
            lens = sorted(set(int(x) for x in rng.randint(m, m + 6, size=nl)))
            mp, ml = [], []
            for L in lens:
                for _ in range(int(rng.randint(1, 40))):
                    if rng.rand() < 0.6:
                        off = int(rng.randint(0, len(text) - L))
                        mp.append(text[off:off + L])
                    else:
                        mp.append(rng.randint(0, sigma, size=L).astype(np.uint8))
                    ml.append(L)
            order = rng.permutation(len(ml))
            mpat = np.concatenate([mp[i] for i in order])
            mlen = np.array([ml[i] for i in order], dtype=np.uint32)
            wantset = 0
            for L in lens:
                flat = np.concatenate([mp[i] for i in order if ml[i] == L])
                wantset += O.count_bruteforce(flat, L, len(flat) // L, text)
            for algo in (S.ALGO_AC, S.ALGO_WM):
                if algo == S.ALGO_WM and sigma not in (2, 4, 8, 20, 128, 256):
                    continue
                ps = S.PatternSet(mpat, mlen, sigma, algo)
                assert ps.count_host(text)[0] == wantset, (tag, "pset", algo, lens, ps.info().one_pass)
                cur = torch.zeros(1, dtype=torch.int64, device=dev)
                out = torch.zeros(wantset + 4, dtype=torch.int64, device=dev)
                ps.positions_device(d_text.data_ptr(), len(text), out.data_ptr(), wantset + 4, cur.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert int(cur.item()) == wantset, (tag, "pset positions", algo, lens)
                checks += 2
                if algo == S.ALGO_WM and sigma == 4 and lens[0] >= 8:
                    # the grouped pair-gram form forced whatever its candidate rate, its verify stage by the suffix index and
                    # class by class: same count, same END columns
                    ref = np.sort(out[:wantset].cpu().numpy())
                    T = S.load_testing()  # the knobs exist only in the testing build (csrc/smh_tune.h)
                    for tune in ("grouped=force", "grouped=force,sfx=0"):
                        T.tune(T.TUNE_WM, tune)  # read by the table builder and by the launcher
                        try:
                            pg = T.PatternSet(mpat, mlen, sigma, algo)
                            assert pg.count_host(text)[0] == wantset, (tag, "pset", tune, lens)
                            cur.zero_()
                            out.zero_()
                            pg.positions_device(d_text.data_ptr(), len(text), out.data_ptr(), wantset + 4, cur.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream)
                            torch.cuda.synchronize()
                        finally:
                            T.tune(T.TUNE_WM, None)
                        assert int(cur.item()) == wantset and np.array_equal(np.sort(out[:wantset].cpu().numpy()), ref), (tag, "pset positions", tune, lens)
                        pg.close()
                        checks += 2
        checks += 8
        if verbose:
            print(tag, "ok", flush=True)
    if verbose:
        print("fuzz: %d cases, %d checks, all equal" % (cases, checks))
    return checks


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
