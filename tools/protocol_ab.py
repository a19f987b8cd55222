#!/usr/bin/env python3
"""Two timing protocols on the SAME handle, text and box, alternated: (A) what the interleaved A/B tools do -- one launch, one
torch.cuda.synchronize(), median of 20 -- and (B) what bench.py does -- 25 ms of back-to-back launches, then 20 back-to-back
launches each between its own events, median.  Says how much of the gap between a notes/ab_*.log figure and the bench record
is the protocol (the device's clocks under sustained load) and how much the box.
usage: protocol_ab.py [m p MiB alphabet]..."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
sets = [tuple(int(x) for x in sys.argv[i:i + 4]) for i in range(1, len(sys.argv) - 3, 4)] or [(12, 100000, 4096, 256), (8, 100000, 4096, 256), (16, 1000, 1024, 4)]
texts = {}


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


for m, p, mib, sigma in sets:
    n = mib << 20
    if (mib, sigma) not in texts:
        texts.clear()
        t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.lib.smh_corpus_text_device(C.c_void_p(t.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
        texts[(mib, sigma)] = t
    text = texts[(mib, sigma)]
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    h = (S.WmTables if sigma > 4 else S.AcAutomaton).from_patterns(pat, m, p, sigma)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    launch = lambda: h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    out = []
    for rnd in range(3):
        a_ms = []
        for _ in range(20):  # (A) isolated launches
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); launch(); b.record()
            torch.cuda.synchronize()
            a_ms.append(a.elapsed_time(b))
        est = median(a_ms)
        for _ in range(int(25.0 / est) + 2):  # (B) sustained
            launch()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record(); launch(); b.record()
        torch.cuda.synchronize()
        b_ms = [a.elapsed_time(b) for a, b in evs]
        out.append((median(a_ms), median(b_ms)))
    print("m=%d p=%d sigma=%d %d MiB: " % (m, p, sigma, mib) + "; ".join("isolated %.4f ms (%.3f) / sustained %.4f ms (%.3f)" % (a, n / a / 8e9, b, n / b / 8e9) for a, b in out)
          + "  [sustained / isolated = %.3f]" % median([b / a for a, b in out]), flush=True)
