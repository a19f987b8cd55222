#!/usr/bin/env python3
"""Development A/B of two BUILDS of the library in one process: the same pattern sets compiled by each, launches interleaved
on the same resident text.  usage: lib_ab.py libA.so libB.so [--mib 1024] [--sets ac:1000:16,ac:1000:32,wm:100000:12:256,...]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402  (corpus only)

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--mib", type=int, default=1024)
ap.add_argument("--sets", default="ac:1000:8,ac:1000:16,ac:1000:32,ac:8000:16")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--engine", type=int, default=-1, help="smh_*_set_scan_engine value forced on both handles (3 = key table, 4 = window hash)")
args = ap.parse_args()


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.smh_ac_compile_patterns.restype = C.c_void_p
    L.smh_ac_compile_patterns.argtypes = [S.u8p, C.c_int, C.c_int, C.c_int]
    L.smh_wm_compile.restype = C.c_void_p
    L.smh_wm_compile.argtypes = [S.u8p, C.c_int, C.c_int, C.c_int]
    for f in (L.smh_ac_scan, L.smh_wm_scan):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    L.smh_last_error.restype = C.c_char_p
    return L


libs = [load(p) for p in args.libs]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
texts = {}
for spec in args.sets.split(","):
    f = spec.split(":")
    algo, p, m = f[0], int(f[1]), int(f[2])
    sigma = int(f[3]) if len(f) > 3 else 4
    mib = int(f[4]) if len(f) > 4 else args.mib
    n = mib << 20
    if (sigma, mib) not in texts:
        texts.clear()
        t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(t.data_ptr(), n, 42, sigma)
        torch.cuda.synchronize()
        texts[(sigma, mib)] = t
    text = texts[(sigma, mib)]
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    hs = []
    for L in libs:
        h = (L.smh_ac_compile_patterns if algo == "ac" else L.smh_wm_compile)(pat.ctypes.data_as(S.u8p), m, p, sigma)
        assert h, L.smh_last_error()
        if args.engine >= 0:
            rc = (L.smh_ac_set_scan_engine if algo == "ac" else L.smh_wm_set_scan_engine)(C.c_void_p(h), args.engine)
            assert rc == 0, L.smh_last_error()
        hs.append(C.c_void_p(h))
    ts = [[], []]
    counts = [None, None]
    for it in range(args.reps + 4):
        for i, L in enumerate(libs):
            scan = L.smh_ac_scan if algo == "ac" else L.smh_wm_scan
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnt.zero_()
            a.record()
            rc = scan(hs[i], C.c_void_p(text.data_ptr()), n, C.c_void_p(cnt.data_ptr()), 0, C.c_void_p(st))
            b.record()
            assert rc == 0, L.smh_last_error()
            torch.cuda.synchronize()
            if it >= 4:
                ts[i].append(a.elapsed_time(b))
            counts[i] = int(cnt.item())
    med = [sorted(t)[len(t) // 2] for t in ts]
    print("%-22s A %.4f ms  B %.4f ms  B/A %.3f  (min %.4f / %.4f) counts %s" % (spec, med[0], med[1], med[1] / med[0], min(ts[0]), min(ts[1]),
                                                                           "equal" if counts[0] == counts[1] else counts), flush=True)
