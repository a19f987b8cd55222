#!/usr/bin/env python3
"""Development micro-driver: time a mixed-length pattern set (BASELINE configs[1] read as ONE set:
1000 DNA patterns with lengths drawn from 8..32) with both algorithms."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
mib, p, lo, hi, sigma = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
host = S.corpus_text(min(n, 1 << 26), 42, sigma)
rng = np.random.RandomState(5)
lengths = rng.randint(lo, hi + 1, size=p).astype(np.uint32)
pats = []
for j, L in enumerate(lengths):
    if j % 2 == 0:
        off = int(rng.randint(0, len(host) - L))
        pats.append(host[off:off + L])
    else:
        pats.append(rng.randint(0, sigma, size=L).astype(np.uint8))
patterns = np.concatenate(pats)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for algo, name in ((S.ALGO_WM, "WM"), (S.ALGO_AC, "AC")):
    ps = S.PatternSet(patterns, lengths, sigma, algo)
    ps.scan_device(text.data_ptr(), n, cnt.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); ps.scan_device(text.data_ptr(), n, cnt.data_ptr(), st); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    i = ps.info()
    print("pset %s sigma=%d p=%d lengths %d..%d (%d classes, one_pass=%d) %d MiB: median %.3f ms %.0f GB/s  count %d"
          % (name, sigma, p, lo, hi, i.classes, i.one_pass, mib, ts[2], n / ts[2] / 1e6, int(cnt.item())))
