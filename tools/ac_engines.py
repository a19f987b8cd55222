#!/usr/bin/env python3
"""Development A/B of the Aho-Corasick entry point's engine choice: the handle as compiled, the automaton kernels forced
(smh_ac_set_scan_engine(SMH_ALGO_AC)) and, where the compile kept one, the filter engine forced; launches interleaved.
usage: ac_engines.py MiB alphabet m:p [m:p ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
mib, sigma = int(sys.argv[1]), int(sys.argv[2])
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, sigma, C.c_void_p(st))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for spec in sys.argv[3:]:
    m, p = (int(x) for x in spec.split(":"))
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    hs = {"default": S.AcAutomaton.from_patterns(pat, m, p, sigma), "automaton": S.AcAutomaton.from_patterns(pat, m, p, sigma)}
    hs["automaton"].set_scan_engine(S.ALGO_AC)
    try:
        h = S.AcAutomaton.from_patterns(pat, m, p, sigma)
        h.set_scan_engine(S.ALGO_WM)
        hs["filter"] = h
    except S.SmhError:
        pass
    ts = {k: [] for k in hs}
    counts = {}
    for it in range(33):
        for k, h in hs.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnt.zero_(); a.record(); h.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts[k].append(a.elapsed_time(b))
            counts[k] = int(cnt.item())
    i = hs["default"].info()
    assert len(set(counts.values())) == 1, counts
    print("m=%d p=%d engine %s stride %d K %d exact %d full %d dense %d:" % (m, p, "filter" if i.scan_engine == S.ALGO_WM else "automaton", i.scan_stride, i.scan_depth, i.scan_exact, i.scan_full_rows, i.scan_dense),
          "  ".join("%s %.4f" % (k, sorted(v)[len(v) // 2]) for k, v in ts.items()), " count", counts["default"])
