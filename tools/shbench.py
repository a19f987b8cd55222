#!/usr/bin/env python3
"""Development micro-driver: time the Set-Horspool paths for one (m, p, MiB, alphabet)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
m, p, mib, sigma = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
sh = S.ShTrie.from_patterns(pat, m, p, sigma)
bm = sh.valid_bmbc()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for variant, name in ((S.VARIANT_TUNED, "tuned"), (S.VARIANT_TABLE, "table walk")):
    sh.scan_device(text.data_ptr(), n, cnt.data_ptr(), bm, variant, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); sh.scan_device(text.data_ptr(), n, cnt.data_ptr(), bm, variant, st); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print("SH sigma=%d m=%d p=%d %d MiB %s (engine %s, mean shift %.2f): median %.4f ms %.0f GB/s  count %d"
          % (sigma, m, p, mib, name, "WM" if sh.info().tuned_engine else "AC", float(bm.mean()), ts[2], n / ts[2] / 1e6, int(cnt.item())))
