#!/usr/bin/env python3
"""Development A/B: the automaton kernels (engine forced) of one AC handle timed under two SMH_AC_TUNE settings, launches interleaved.
usage: ac_ab.py m p MiB tuneA tuneB"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
S = S.load_testing()  # knobs exist only in the testing twin (csrc/smh_tune.h)
m, p, mib = (int(x) for x in sys.argv[1:4])
tunes = sys.argv[4:6]
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, 4, 42, n, 2)
ac = S.AcAutomaton.from_patterns(pat, m, p, 4)
ac.set_scan_engine(S.ALGO_AC)
i = ac.info()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ts = {t: [] for t in tunes}
counts = {}
for it in range(43):
    for t in tunes:
        S.tune(S.TUNE_AC, t)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts[t].append(a.elapsed_time(b))
        counts[t] = int(cnt.item())
for t in tunes:
    v = sorted(ts[t])
    print("m=%d p=%d %d MiB stride=%d K=%d exact=%d full_rows=%d tune=%-10s median %.4f ms (%.3f of 8 TB/s)  min %.4f  mean %.4f  count %d"
          % (m, p, mib, i.scan_stride, i.scan_depth, i.scan_exact, i.scan_full_rows, t, v[len(v) // 2], n / v[len(v) // 2] / 1e6 / 8000, v[0], sum(v) / len(v), counts[t]))
