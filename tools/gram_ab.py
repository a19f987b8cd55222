#!/usr/bin/env python3
"""Development A/B: one Wu-Manber set compiled under two (or more) SMH_WM_TUNE settings -- e.g. "gram=2" and "gram=8", the hashed byte-gram
filter in its 128 KiB and its 143.9 KiB table -- scanned over the same text, launches interleaved in one process; each also with the
survivors dropped ("stmin=-1": the filter alone, counts wrong).  Testing twin: the knobs exist only there.
usage: gram_ab.py m p MiB alphabet tuneA tuneB [...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
T = S.load_testing()
m, p, mib, sigma = (int(x) for x in sys.argv[1:5])
tunes = sys.argv[5:]
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
T.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
pat = T.corpus_patterns(m, p, 7, sigma, 42, n, 2)
hs = {}
for t in tunes:
    T.tune(T.TUNE_WM, t)
    hs[t] = T.WmTables.from_patterns(pat, m, p, sigma)
    if hs[t].info().scan_engine != T.ALGO_WM:
        hs[t].set_scan_engine(T.ALGO_WM)
T.tune(T.TUNE_WM, None)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
variants = [(t, "") for t in tunes] + [(t, "stmin=-1") for t in tunes]
ts = {v: [] for v in variants}
counts = {}
for it in range(23):
    for v in variants:
        T.tune(T.TUNE_WM, v[1] or None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); hs[v[0]].scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts[v].append(a.elapsed_time(b))
        counts[v] = int(cnt.item())
T.tune(T.TUNE_WM, None)
for v in variants:
    x = sorted(ts[v]); i = hs[v[0]].info(); ad = hs[v[0]].adapt()
    print("m=%d p=%d sigma=%d %d MiB compiled %-8s form %d, %6d B LDS%s: median %.4f ms = %.3f of 8 TB/s (min %.4f, mean %.4f) survivors/4KiB %.1f count %d"
          % (m, p, sigma, mib, v[0], i.gram_kind, i.lds_bytes, "  FILTER ALONE (counts wrong)" if v[1] else "", x[len(x) // 2], n / x[len(x) // 2] / 1e6 / 8000, x[0], sum(x) / len(x),
             ad.events_per_4k[T.ALGO_WM], counts[v]), flush=True)
