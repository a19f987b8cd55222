#!/usr/bin/env python3
"""Development A/B: the same WM handle timed with two SMH_WM_TUNE settings, launches interleaved in one process.
usage: wm_ab.py m p MiB alphabet tuneA tuneB"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
S = S.load_testing()  # knobs exist only in the testing twin (csrc/smh_tune.h)
m, p, mib, sigma = (int(x) for x in sys.argv[1:5])
tunes = sys.argv[5:7]
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
wm = S.WmTables.from_patterns(pat, m, p, sigma)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ts = {t: [] for t in tunes}
counts = {}
for it in range(43):
    for t in tunes:
        S.tune(S.TUNE_WM, t)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts[t].append(a.elapsed_time(b))
        counts[t] = int(cnt.item())
for t in tunes:
    v = sorted(ts[t])
    print("   in order:", " ".join("%.0f" % (x * 1000) for x in ts[t]))
    print("m=%d p=%d %d MiB tune=%-10s median %.4f ms  min %.4f  mean %.4f  count %d" % (m, p, mib, t, v[len(v) // 2], v[0], sum(v) / len(v), counts[t]))
