"""Development aid: the compile's estimates (smh_adapt_info.est_ms_per_gib) against measured rates on uniform text, per engine,
for a grid of sets.  usage: est_check.py SIGMA [P,P,...] [M,M,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch  # noqa: E402
import smatcher_hip as S  # noqa: E402

sigma = int(sys.argv[1])
ps = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "300,1000,3000").split(",")]
ms = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,12,16,24").split(",")]
n = 1 << 30
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.corpus_text_device(text.data_ptr(), n, 42, sigma)
torch.cuda.synchronize()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(h, reps=10):
    out = []
    for _ in range(reps):
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return sorted(out[2:])[len(out[2:]) // 2]


for make, name in ((S.AcAutomaton, "ac"), (S.WmTables, "wm")):
    for p in ps:
        for m in ms:
            pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
            try:
                h = make.from_patterns(pat, m, p, sigma)
            except S.SmhError as e:
                print(name, p, m, "n/a", str(e)[:40])
                continue
            est = list(h.adapt().est_ms_per_gib)
            row = []
            for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT):
                try:
                    h.set_scan_engine(eng)
                except S.SmhError:
                    row.append(None)
                    continue
                row.append(round(timed(h), 4))
            h.close()
            print("%s sigma=%d p=%d m=%d  est %s  measured %s  ratio %s" % (name, sigma, p, m, [round(x, 3) for x in est], row,
                  [round(r / e, 2) if r and e > 0 else None for r, e in zip(row, est)]), flush=True)
