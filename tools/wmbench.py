#!/usr/bin/env python3
"""Development micro-driver: time the WM kernels for one (m, p, MiB, alphabet)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
S = S.for_tools()  # knobs exist only in the testing twin (csrc/smh_tune.h)
m, p, mib, sigma = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
every = int(sys.argv[6]) if len(sys.argv) > 6 else 2  # every `every`-th pattern is cut from the text (0: none)
pat = S.corpus_patterns(m, p, 7, sigma, 42, n, every)
wm = S.WmTables.from_patterns(pat, m, p, sigma)
if "gram=" in os.environ.get("SMH_WM_TUNE", "") and wm.info().scan_engine != S.ALGO_WM:
    wm.set_scan_engine(S.ALGO_WM)  # time this path's own kernels, not the automaton engine
i = wm.info()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(2):
    wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), variant, st)
torch.cuda.synchronize()
ts = []
for _ in range(21):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cnt.zero_(); a.record(); wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), variant, st); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("WM sigma=%d m=%d p=%d %d MiB variant=%d W=%d T=%d exact=%d hashed=%d planes=%d engine=%d: median %.4f ms %.0f GB/s  min %.4f  count %d"
      % (sigma, m, p, mib, variant, i.block_symbols, i.filter_log2, i.filter_exact, i.filter_hashed, i.gram_planes, i.scan_engine, ts[10], n / ts[10] / 1e6, ts[0], int(cnt.item())))
