#!/usr/bin/env python3
"""What the launches of a Wu-Manber handle REPORT about the text beside what its compile simulated: surviving columns per 4 KiB
chunk (smh_adapt_info.events_per_4k against smh_wm_info: the builder's pseudo-random text), and which verify entries the
launch therefore reads.  usage: survivors.py [m ...]   (100 000 byte patterns, 1 GiB)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
n = 1 << 30
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 256, C.c_void_p(st))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for m in [int(x) for x in sys.argv[1:]] or [5, 6, 7, 8, 9, 12]:
    pat = S.corpus_patterns(m, 100000, 7, 256, 42, n, 2)
    h = S.WmTables.from_patterns(pat, m, 100000, 256)
    seen = []
    for _ in range(12):
        h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
        torch.cuda.synchronize()
        ad = h.adapt()
        seen.append("%.1f/%.1f" % (ad.events_per_4k[S.ALGO_WM], 4096.0 * ad.verify_density))
    print("m=%d after each launch, reported / what the next launch plans for: %s" % (m, " ".join(seen)))
    i, ad = h.info(), h.adapt()
    print("m=%d form %d planes %d: engine %d, reports %d, measured %.1f surviving columns per 4 KiB, ms/GiB measured %s estimated %s"
          % (m, i.gram_kind, i.gram_planes, ad.engine, ad.reports, ad.events_per_4k[S.ALGO_WM],
             ["%.3f" % x for x in ad.ms_per_gib], ["%.3f" % x for x in ad.est_ms_per_gib]), flush=True)
