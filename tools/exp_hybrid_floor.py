#!/usr/bin/env python3
"""Development experiment (round 3): what would the m=16 automaton scan cost without compact-row resolutions,
and what does a candidate cost?  (a) the hybrid K=12/D=9 image on a text that can never drive a lane deeper
than 9 (patterns carry symbol 3 at position 9, the text has no 3); (b) the plain stride-2 image cut at
K = 9 / 8 (no compact rows; candidate rates 1000/4^9, 1000/4^8) on the normal text; (c) the normal plan."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch
import smatcher_hip as S
n = 1 << 30
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def text_of(sigma):
    t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    S.lib.smh_corpus_text_device(C.c_void_p(t.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
    return t


def run(label, ac, text):
    for _ in range(3):
        ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    i = ac.info()
    print("%-46s stride %d K %d full_rows %d lds %d KB: median %.4f ms (%.0f GB/s) min %.4f count %d"
          % (label, i.scan_stride, i.scan_depth, i.scan_full_rows, i.lds_bytes >> 10, ts[6], n / ts[6] / 1e6, ts[0], int(cnt.item())), flush=True)


m, p = 16, 1000
t4, t3 = text_of(4), text_of(3)
pat = S.corpus_patterns(m, p, 7, 4, 42, n, 2)
ac = S.AcAutomaton.from_patterns(pat, m, p, 4)
run("(c) normal plan, normal text", ac, t4)
run("(c') normal plan, text without symbol 3", ac, t3)
pat9 = pat.copy().reshape(p, m)
pat9[:, 9] = 3
pat9 = np.ascontiguousarray(pat9.reshape(-1))
ac9 = S.AcAutomaton.from_patterns(pat9, m, p, 4)
run("(a) symbol 3 at position 9, text without 3", ac9, t3)
pat7 = pat.copy().reshape(p, m)
pat7[:, 7] = 3
ac7 = S.AcAutomaton.from_patterns(np.ascontiguousarray(pat7.reshape(-1)), m, p, 4)
run("(a') symbol 3 at position 7, text without 3", ac7, t3)
for K in (9, 8, 7):
    ac.set_scan_plan(2, K)
    run("(b) plain stride 2 cut at K=%d" % K, ac, t4)
ac.set_scan_plan(3, 12 | (8 << 8))
run("(d) hybrid K=12 D=8", ac, t4)
