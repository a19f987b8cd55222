#!/usr/bin/env python3
"""DNA sets x the three pair-like gram forms (testing twin, "gram=1|3|5"), the handle pinned to its filter kernels (no adaptive
switching), 1 GiB: median scan time beside the survivors per chunk the compile simulated -- what wm_host.c's
gram_verify_ms_pairlike is fitted to.  Bare scans of the forms (no survivors): pair form 0.173, two-column 8-grams 0.173, 8-grams
0.215 ms/GiB.  usage: l2_fit.py [l2=0|l2=1|regv=1|""] [m:p,m:p,...]"""
import ctypes as C, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S
T = S.load_testing()
extra = sys.argv[1] if len(sys.argv) > 1 else ""
n = 1 << 30
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
T.lib.smh_corpus_text_device(C.c_void_p(text.data_ptr()), n, 0, 42, 4, C.c_void_p(st))
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def compile_with_debug(pat, m, p, tune):
    """-> (handle, builder's stderr)"""
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+") as tmp:
        saved = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            T.tune(T.TUNE_WM, tune + " debug")
            h = T.WmTables.from_patterns(pat, m, p, 4)
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return h, tmp.read()


SETS = [(10, 2000), (12, 2000), (12, 4000), (12, 8000), (13, 8000), (14, 8000), (14, 20000), (16, 8000), (16, 20000), (16, 40000), (18, 40000), (20, 40000), (24, 40000)]
if len(sys.argv) > 2:  # "16:2000,16:4000"
    SETS = [tuple(int(x) for x in t.split(":")) for t in sys.argv[2].split(",")]
for m, p in SETS:
    pat = T.corpus_patterns(m, p, 7, 4, 42, n, 2)
    for g in (1, 3, 5):
        h, err = compile_with_debug(pat, m, p, "gram=%d" % g)
        mt = re.search(r"kept form (\d+), (\d+) planes, survivors ([0-9.]+), est ([0-9.]+)", err)
        i = h.info()
        if i.gram_kind != g or not mt:
            continue
        h.set_scan_engine(T.ALGO_WM)
        T.tune(T.TUNE_WM, extra or None)
        ts = []
        for it in range(14):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnt.zero_(); a.record(); h.scan_device(text.data_ptr(), n, cnt.data_ptr(), T.VARIANT_TUNED, st); b.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(a.elapsed_time(b))
        ts.sort()
        print("m=%d p=%d form %d planes %s: simulated %.2f survivors per 4 KiB, model %.3f ms/GiB, measured median %.4f (min %.4f) [%s] count %d"
              % (m, p, g, mt.group(2), 4096 * float(mt.group(3)), float(mt.group(4)), ts[len(ts) // 2], ts[0], extra or "default", int(cnt.item())), flush=True)
T.tune(T.TUNE_WM, None)
