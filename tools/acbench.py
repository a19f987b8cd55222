#!/usr/bin/env python3
"""Development micro-driver: time the tuned AC kernel for one (m, p, MiB) under the SMH_AC_TUNE knobs."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch
import smatcher_hip as S
S = S.for_tools()  # knobs exist only in the testing twin (csrc/smh_tune.h)
m, p, mib = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
stride, depth = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, 0)
n = mib << 20
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
S.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
pat = S.corpus_patterns(m, p, 7, 4, 42, n, 2)
ac = S.AcAutomaton.from_patterns(pat, m, p, 4)
if stride or depth:
    ac.set_scan_plan(stride, depth)
i = ac.info()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(3):
    ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cnt.zero_(); a.record(); ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
    torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print("m=%d p=%d %d MiB tune=%s stride=%d K=%d exact=%d lds=%dKB: median %.4f ms %.0f GB/s  min %.4f ms %.0f GB/s  count %d"
      % (m, p, mib, os.environ.get("SMH_AC_TUNE", "-"), i.scan_stride, i.scan_depth, i.scan_exact, i.lds_bytes >> 10,
         ts[5], n / ts[5] / 1e6, ts[0], n / ts[0] / 1e6, int(cnt.item())))
