#!/usr/bin/env python3
"""Development A/B: the key engine's two images -- two-table cuckoo (layout 0) and bucket image (layout 1, csrc/key_hash.h) -- of the
same set over the same text, launches interleaved in one process (testing twin: the layout is a development knob).
usage: key_ab.py [MiB]"""
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
CORP = [("uniform", T.CORPUS_UNIFORM), ("repeats", T.CORPUS_DNA_REPEATS), ("skewed", T.CORPUS_SKEWED)]
SETS = [(4, 16, 8000), (4, 16, 3000), (4, 20, 8000), (20, 8, 10000)] if not os.environ.get('KEY_AB_K14') else [(4, 16, 1500), (4, 16, 2500), (4, 16, 3500), (4, 16, 5000), (20, 8, 3000)]
for sigma, m, p in SETS:
    for cname, kind in CORP:
        if (kind == T.CORPUS_DNA_REPEATS and sigma != 4) or (kind == T.CORPUS_SKEWED and sigma == 4):
            continue
        text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        T.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
        torch.cuda.synchronize()
        pat = T.corpus_patterns(m, p, 12, sigma, 42, n, 2, kind)
        hs = {}
        for layout in (0, 1):
            T.tune(T.TUNE_KEY, "layout=%d%s" % (layout, ",buckets_log2=14" if os.environ.get("KEY_AB_K14") else ""))
            try:
                hs[layout] = T.KeyTable(pat, m, p, sigma)
            except T.SmhError as e:
                print("sigma=%d m=%d p=%d: layout %d not taken (%s)" % (sigma, m, p, layout, e))
        T.tune(T.TUNE_KEY, None)
        if 1 in hs:
            hs["1 no overflow path (counts wrong)"] = hs[1]
        ts = {l: [] for l in hs}
        counts = {}
        for it in range(23):
            for l, k in hs.items():
                T.tune(T.TUNE_KEY, "noover=1" if isinstance(l, str) else None)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                cnt.zero_(); a.record(); k.scan_device(text.data_ptr(), n, cnt.data_ptr(), st); b.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts[l].append(a.elapsed_time(b))
                counts[l] = int(cnt.item())
        T.tune(T.TUNE_KEY, None)
        for l, k in hs.items():
            v = sorted(ts[l]); i = k.info()
            print("sigma=%d m=%d p=%d %-8s layout %s: %5d keys, %6d B LDS, overflow %4d: median %.4f ms / %d MiB = %.3f of 8 TB/s  (min %.4f)  count %d%s"
                  % (sigma, m, p, cname, l, i.keys, i.lds_bytes, i.overflow_keys, v[len(v) // 2], mib, n / v[len(v) // 2] / 1e6 / 8000, v[0], counts[l],
                     "" if counts[0] == counts[1] else "   COUNTS DIFFER"), flush=True)
        for l, k in hs.items():
            if not isinstance(l, str):
                k.close()
        del text
