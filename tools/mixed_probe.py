"""Development aid: the mixed-length set of the bench (1000 patterns, 40 of each length 8..32) through smh_pset_*, per
SMH_WM_TUNE setting given on the command line (the grouped pair-gram form is chosen at compile time).
    python tools/mixed_probe.py "" "grouped=force" ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import smatcher_hip as S  # noqa: E402

n = 1 << 30
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.corpus_text_device(text.data_ptr(), n, 42, 4)
torch.cuda.synchronize()
lo, hi, per = 8, 32, 40
if os.environ.get("MIXED_SHAPE"):
    lo, hi, per = [int(x) for x in os.environ["MIXED_SHAPE"].split(",")]
mlens, mpats = [], []
for L in range(lo, hi + 1):
    mpats.append(S.corpus_patterns(L, per, 7 + 100 + L, 4, 42, n, 2))
    mlens += [L] * per
pats, lens = np.concatenate(mpats), np.array(mlens, dtype=np.uint32)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for tune in sys.argv[1:] or [""]:
    os.environ["SMH_WM_TUNE"] = tune
    for name, algo in (("ac", S.ALGO_AC), ("wm", S.ALGO_WM)):
        ps = S.PatternSet(pats, lens, 4, algo)
        ts = []
        for it in range(24):
            cnt.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ps.scan_device(text.data_ptr(), n, cnt.data_ptr(), stream)
            b.record()
            torch.cuda.synchronize()
            if it >= 4:
                ts.append(a.elapsed_time(b))
        i = ps.info()
        print("tune=%-16r %s: median %.4f ms min %.4f  matches %d one_pass %d classes %d" % (tune, name, sorted(ts)[len(ts) // 2], min(ts), int(cnt.item()),
                                                                                        i.one_pass, i.classes), flush=True)
        ps.close()
