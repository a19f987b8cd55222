#!/usr/bin/env python3
"""Development A/B: one pattern set compiled once per SMH_WM_TUNE setting (the gram FORM is a compile-time choice of the
handle: gram=1 pair form, gram=3 8-symbol grams, gram=5 8-symbol grams at two columns per lookup), launches interleaved.
usage: wm_forms.py m p MiB alphabet tune [tune ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
m, p, mib, sigma = (int(x) for x in sys.argv[1:5])
tunes = sys.argv[5:]
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
wms = {}
for t in tunes:
    os.environ["SMH_WM_TUNE"] = t
    wms[t] = S.WmTables.from_patterns(pat, m, p, sigma)
    if wms[t].info().scan_engine != S.ALGO_WM:
        wms[t].set_scan_engine(S.ALGO_WM)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ts = {t: [] for t in tunes}
counts = {}
for it in range(43):
    for t in tunes:
        os.environ["SMH_WM_TUNE"] = t
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); wms[t].scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts[t].append(a.elapsed_time(b))
        counts[t] = int(cnt.item())
for t in tunes:
    v = sorted(ts[t]); i = wms[t].info()
    print("m=%d p=%d %d MiB tune=%-14s planes %2d regv %d  median %.4f ms  min %.4f  count %d" % (m, p, mib, t, i.gram_planes, i.verify_in_registers, v[len(v) // 2], v[0], counts[t]))
