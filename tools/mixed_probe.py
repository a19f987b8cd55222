import ctypes as C, os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
import torch, numpy as np
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
T = S.load_testing()
n = 1 << 30
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
T.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
pats, lens = [], []
for L in range(8, 33):
    pats.append(T.corpus_patterns(L, 40, 7 + 100 + L, 4, 42, n, 2)); lens += [L] * 40
ps = T.PatternSet(np.concatenate(pats), np.array(lens, dtype=np.uint32), 4, T.ALGO_WM)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for tune in (None, "stmin=-1", None, "stmin=-1"):
    T.tune(T.TUNE_WM, tune)
    ts = []
    for it in range(14):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); ps.scan_device(text.data_ptr(), n, cnt.data_ptr(), st); b.record(); torch.cuda.synchronize()
        if it >= 2: ts.append(a.elapsed_time(b))
    ts.sort(); print(tune, "median %.4f min %.4f count %d" % (ts[len(ts)//2], ts[0], int(cnt.item())), flush=True)
