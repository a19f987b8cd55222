#!/usr/bin/env python3
"""Development micro-driver: the window-hash engine forced inside a Wu-Manber handle, 1 GiB of a corpus (SMH_HASH_TUNE=drop=1: stage 1 alone)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
S = S.for_tools()  # knobs exist only in the testing twin (csrc/smh_tune.h)
m, p, mib, sigma, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = mib << 20
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
pat = S.corpus_patterns(m, p, 12, sigma, 42, n, 2, kind)
wm = S.WmTables.from_patterns(pat, m, p, sigma)
wm.set_scan_engine(S.ENGINE_HASH)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(3):
    wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
torch.cuda.synchronize()
ts = []
for _ in range(11):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cnt.zero_(); a.record(); wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("hash sigma=%d m=%d p=%d kind=%d %s: median %.4f ms/%d MiB (%.3f of 8 TB/s) min %.4f count %d events/4k %.1f"
      % (sigma, m, p, kind, os.environ.get("SMH_HASH_TUNE", ""), ts[5], mib, n / ts[5] / 1e6 / 8000, ts[0], int(cnt.item()), wm.adapt().events_per_4k[S.ENGINE_HASH]))
