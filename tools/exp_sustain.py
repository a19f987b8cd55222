#!/usr/bin/env python3
"""Experiment (round 3): why the headline kernels take 0.19-0.20 ms inside bench.py's step loop and 0.164-0.174 ms when timed
alone.  Prints per-launch event times of (a) one kernel repeated back to back, (b) the three kernels alternating (the step),
(c) the same with an idle gap in front, so a clock ramp-down under sustained load would show as a time series."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
n = 1 << 30
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
acs = {m: S.AcAutomaton.from_patterns(S.corpus_patterns(m, 1000, 7, 4, 42, n, 2), m, 1000, 4) for m in (8, 16, 32)}
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
for m in acs:
    acs[m].scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)

def series(order, reps, label):
    evs = [ev() for _ in range(reps * len(order) + 1)]
    evs[0].record()
    k = 1
    for _ in range(reps):
        for m in order:
            acs[m].scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
            evs[k].record(); k += 1
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) * 1000 for i in range(len(evs) - 1)]
    print(label, "us:", " ".join("%.0f" % t for t in ts))
    return ts

for m in (8, 16, 32):
    time.sleep(0.5)
    series((m,), 60, "m=%d alone x60" % m)
time.sleep(0.5)
series((8, 16, 32), 30, "step x30")
time.sleep(0.5)
series((8, 16, 32), 30, "step x30 again")
series((8, 16, 32), 100, "step x100 no idle gap")
