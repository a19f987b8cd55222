#!/usr/bin/env python3
"""Development A/B: the window-hash engine with two and with three filter bits per window (csrc/hash_engine.h bloom_k), the same set over
the same text, launches interleaved in one process (testing twin: the bit count is a development knob).  usage: hash_ab.py [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
T = S.load_testing()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
CASES = [(256, 8, 100000, T.CORPUS_SKEWED), (256, 12, 100000, T.CORPUS_SKEWED), (256, 20, 100000, T.CORPUS_SKEWED), (256, 12, 100000, T.CORPUS_UNIFORM),
         (20, 16, 1000, T.CORPUS_SKEWED), (256, 8, 30000, T.CORPUS_SKEWED)]
for sigma, m, p, kind in CASES:
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    T.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
    torch.cuda.synchronize()
    pat = T.corpus_patterns(m, p, 12, sigma, 42, n, 2, kind)
    hs = {}
    for bits in (2, 3):
        T.tune(T.TUNE_HASH, "bits=%d" % bits)
        h = (T.AcAutomaton if sigma == 20 else T.WmTables).from_patterns(pat, m, p, sigma)
        h.set_scan_engine(T.ENGINE_HASH)
        hs[bits] = h
    T.tune(T.TUNE_HASH, None)
    variants = [(b, False) for b in hs] + [(b, True) for b in hs]  # (bits, stage 1 alone: candidates not verified, counts wrong)
    ts = {v: [] for v in variants}
    counts = {}
    for it in range(15):
        for v in variants:
            T.tune(T.TUNE_HASH, "drop=1" if v[1] else None)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnt.zero_(); a.record(); hs[v[0]].scan_device(text.data_ptr(), n, cnt.data_ptr(), T.VARIANT_TUNED, st); b.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts[v].append(a.elapsed_time(b))
            counts[v] = int(cnt.item())
    T.tune(T.TUNE_HASH, None)
    for v in variants:
        x = sorted(ts[v])
        print("sigma=%d m=%d p=%d %-8s %d filter bits%s: median %.4f ms / %d MiB = %.3f of 8 TB/s (min %.4f) candidates/4KiB %.1f count %d%s"
              % (sigma, m, p, T.CORPUS_NAMES[kind], v[0], " stage 1 alone (counts wrong)" if v[1] else "", x[len(x) // 2], mib, n / x[len(x) // 2] / 1e6 / 8000, x[0],
                 hs[v[0]].adapt().events_per_4k[T.ENGINE_HASH], counts[v], "" if v[1] or counts[(2, False)] == counts[(3, False)] else "  COUNTS DIFFER"), flush=True)
    for h in hs.values():
        h.close()
    del text
