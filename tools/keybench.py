#!/usr/bin/env python3
"""Development micro-driver: the key engine (csrc/key_kernels.hip) on 1 GiB of each corpus, against the handle's own engines' count."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
S = S.for_tools()  # knobs exist only in the testing twin (csrc/smh_tune.h)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(launch, reps=9):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); launch(); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


CORP = [("uniform", S.CORPUS_UNIFORM), ("repeats", S.CORPUS_DNA_REPEATS), ("planted", S.CORPUS_PLANTED), ("skewed", S.CORPUS_SKEWED)]
SETS = [(4, 16, 8000), (4, 32, 8000), (4, 20, 8000), (20, 8, 10000), (20, 12, 5000), (256, 8, 5000), (256, 5, 10000)]
for sigma, m, p in SETS:
    for cname, kind in CORP:
        if (kind == S.CORPUS_DNA_REPEATS and sigma != 4) or (kind == S.CORPUS_SKEWED and sigma == 4):
            continue
        text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, st)
        torch.cuda.synchronize()
        pat = S.corpus_patterns(m, p, 12, sigma, 42, n, 2, kind)
        k = S.KeyTable(pat, m, p, sigma)
        i = k.info()
        med, mn = timed(lambda: k.scan_device(text.data_ptr(), n, cnt.data_ptr(), st))
        got = int(cnt.item())
        # the same count by an engine that existed before: a (shorter) prefix through the Aho-Corasick entry point
        chk = min(n, 128 << 20)
        cnt.zero_(); k.scan_device(text.data_ptr(), chk, cnt.data_ptr(), st); torch.cuda.synchronize(); got_chk = int(cnt.item())
        ac = S.AcAutomaton.from_patterns(pat, m, p, sigma) if sigma <= 32 else S.WmTables.from_patterns(pat, m, p, sigma)
        cnt.zero_(); ac.scan_device(text.data_ptr(), chk, cnt.data_ptr(), S.VARIANT_TUNED, st); torch.cuda.synchronize(); want = int(cnt.item())
        print("keys sigma=%d m=%d p=%d %s: %d keys of %d bits, %d slots x2, %d B LDS: median %.4f ms/%d MiB = %.0f GB/s (%.3f of 8 TB/s) min %.4f  count %d  prefix %d %s"
              % (sigma, m, p, cname, i.keys, i.key_bits, i.slots, i.lds_bytes, med, mib, n / med / 1e6, n / med / 1e6 / 8000, mn, got, got_chk,
                 "== other engine" if got_chk == want else "!= OTHER ENGINE %d" % want), flush=True)
        k.close(); ac.close(); del text
