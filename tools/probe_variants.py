#!/usr/bin/env python3
"""time every variant of smh_stream_read_probe_variant on 1 GiB and 4 GiB buffers"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch
import smatcher_hip as S
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(1, dtype=torch.int64, device=dev)
for gib in (1, 4):
    n = gib << 30
    t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    S.lib.smh_corpus_text_device(C.c_void_p(t.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
    for v in range(7):
        for _ in range(2):
            S.lib.smh_stream_read_probe_variant(C.c_void_p(t.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(st), v)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); S.lib.smh_stream_read_probe_variant(C.c_void_p(t.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(st), v); b.record()
            torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort()
        print("%d GiB variant %d: median %.4f ms %.0f GB/s" % (gib, v, ts[4], n / ts[4] / 1e6), flush=True)
    del t
