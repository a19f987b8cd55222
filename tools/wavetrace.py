#!/usr/bin/env python3
"""Development aid: where a tuned AC launch spends its wall time -- per-wave timestamps.

    python tools/wavetrace.py M P MIB [stride depth]

The kernel stores, per wave, three 100 MHz timestamps (start, LDS table staged, done).  Prints the launch
span, the start skew over the grid, the staging time and the spread of finishing times (the tail)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import numpy as np
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
nwaves = 16 * 2 * 256
trace = torch.zeros(3 * nwaves, dtype=torch.int64, device=dev)
for _ in range(3):
    ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st)
torch.cuda.synchronize()
S.lib.smh_dev_set_wave_trace.argtypes = [C.c_void_p]
S.lib.smh_dev_set_wave_trace.restype = None
S.lib.smh_dev_set_wave_trace(C.c_void_p(trace.data_ptr()))
for rep in range(3):
    trace.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ac.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record()
    torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(-1, 3).astype(np.int64)
    full = t.copy()
    t = t[t[:, 2] != 0]
    t0 = t[:, 0].min()
    start, staged, done = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0  # us
    q = lambda x: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max())
    print("m=%d stride=%d K=%d lds=%dKB tune=%s: event %.1f us, %d waves, span %.1f us" % (
        m, i.scan_stride, i.scan_depth, i.lds_bytes >> 10, os.environ.get("SMH_AC_TUNE", "-"), a.elapsed_time(b) * 1e3, len(t), done.max()))
    print("   wave start  (us): " + q(start))
    print("   staging     (us): " + q(staged - start))
    print("   scanning    (us): " + q(done - staged))
    print("   wave done   (us): " + q(done))
    # per workgroup (16 waves): when its first / last wave finished -- is the tail inside workgroups or across them?
    wg = full[: (len(full) // 16) * 16].reshape(-1, 16, 3)
    wg = wg[(wg[:, :, 2] != 0).all(axis=1)]
    wmin, wmax = (wg[:, :, 2].min(axis=1) - t0) / 100.0, (wg[:, :, 2].max(axis=1) - t0) / 100.0
    print("   per-WG first wave done: " + q(wmin))
    print("   per-WG last wave done : " + q(wmax))
S.lib.smh_dev_set_wave_trace(None)
