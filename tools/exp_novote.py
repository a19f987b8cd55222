#!/usr/bin/env python3
"""round-3 experiment: the hybrid kernel on a text that never enters a compact row (see exp_hybrid_floor.py), for builds
with and without the per-step vote (-DSMH_EXP_NOVOTE: counts are meaningless)."""
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
t3 = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.lib.smh_corpus_text_device(C.c_void_p(t3.data_ptr()), n, 0, 42, 3, C.c_void_p(st))
for m, pos in ((12, 9), (16, 9)):
    pat = S.corpus_patterns(m, 1000, 7, 4, 42, n, 2).reshape(1000, m).copy()
    pat[:, pos] = 3
    ac = S.AcAutomaton.from_patterns(np.ascontiguousarray(pat.reshape(-1)), m, 1000, 4)
    for _ in range(3):
        ac.scan_device(t3.data_ptr(), n, cnt.data_ptr(), 0, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); ac.scan_device(t3.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    i = ac.info()
    print("m=%d tune=%s K=%d exact=%d full_rows=%d: median %.4f ms min %.4f count %d" % (m, os.environ.get("SMH_AC_TUNE", "-"), i.scan_depth, i.scan_exact, i.scan_full_rows, ts[6], ts[0], int(cnt.item())), flush=True)
