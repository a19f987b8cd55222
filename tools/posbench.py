#!/usr/bin/env python3
"""Development micro-driver: time the match-position kernels for one (algo, m, p, MiB, alphabet)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
algo, m, p, mib, sigma = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
h = S.AcAutomaton.from_patterns(pat, m, p, sigma) if algo == "ac" else S.WmTables.from_patterns(pat, m, p, sigma)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
h.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
torch.cuda.synchronize()
total = int(cnt[0].item())
cap = total + 16
pos = torch.zeros(cap, dtype=torch.int64, device=dev)
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cnt.zero_(); a.record(); h.positions_device(text.data_ptr(), n, pos.data_ptr(), cap, cnt.data_ptr() + 8, st); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("%s positions sigma=%d m=%d p=%d %d MiB: %d matches, median %.3f ms %.0f GB/s (cursor %d)"
      % (algo, sigma, m, p, mib, total, ts[1], n / ts[1] / 1e6, int(cnt[1].item())))
