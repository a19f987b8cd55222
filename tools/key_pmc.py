#!/usr/bin/env python3
"""Development: ONE variant of the key engine launched a few times over 1 GiB of the repeat-rich DNA corpus (8000 patterns of 16), for
a rocprofv3 --pmc pass per variant.  usage: key_pmc.py cuckoo|bucket|bucket_noover [sigma m p]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
T = S.load_testing()
variant = sys.argv[1]
sigma, m, p = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 16, 8000)
n = 1 << 30
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
kind = T.CORPUS_DNA_REPEATS if sigma == 4 else T.CORPUS_SKEWED
T.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
torch.cuda.synchronize()
pat = T.corpus_patterns(m, p, 12, sigma, 42, n, 2, kind)
T.tune(T.TUNE_KEY, "layout=%d" % (0 if variant == "cuckoo" else 1))
k = T.KeyTable(pat, m, p, sigma)
T.tune(T.TUNE_KEY, "noover=1" if variant == "bucket_noover" else None)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(6):
    cnt.zero_()
    k.scan_device(text.data_ptr(), n, cnt.data_ptr(), st)
torch.cuda.synchronize()
print(variant, "layout", k.info().layout, "count", int(cnt.item()))
