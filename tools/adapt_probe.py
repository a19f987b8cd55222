"""Development aid (round 4): every BASELINE pattern shape on every corpus kind -- the entry point as compiled (the
adaptive engine after a few launches), the automaton kernels forced, the filter kernels forced; counts must agree.

    python tools/adapt_probe.py [--mib 1024] [--sets ac1000,ac8000,wm10000,wm_ascii] [--kinds 0,1,2,3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch  # noqa: E402
import smatcher_hip as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=1024)
ap.add_argument("--sets", default="ac1000,ac8000,wm_long")
ap.add_argument("--kinds", default="0,1,3")
ap.add_argument("--reps", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0)
n = args.mib << 20
stream = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(h, text, reps):
    out = []
    for _ in range(reps):
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, stream)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return out, int(cnt.item())


SETS = {"ac1000": ("ac", 4, 1000, (8, 16, 32)), "ac8000": ("ac", 4, 8000, (16, 32)), "wm_long": ("wm", 4, 1000, (16, 32)),
        "wm_ascii": ("wm", 256, 100000, (8, 12, 20)), "protein": ("ac", 20, 1000, (8, 16)), "wm8000": ("wm", 4, 8000, (16, 32)),
        "wm_protein": ("wm", 20, 1000, (8, 16)), "wm10000": ("wm", 4, 10000, (8, 12))}
for kind in [int(k) for k in args.kinds.split(",")]:
    for name in args.sets.split(","):
        algo, sigma, p, lengths = SETS[name]
        if kind == 1 and sigma != 4:
            continue
        text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(text.data_ptr(), n, 42, sigma, 0, kind, stream)
        torch.cuda.synchronize()
        for m in lengths:
            pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, kind)
            make = S.AcAutomaton if algo == "ac" else S.WmTables
            rec = {"corpus": S.CORPUS_NAMES[kind], "set": name, "m": m}
            h = make.from_patterns(pat, m, p, sigma)
            info = h.info()
            rec["compiled_engine"], rec["adaptive"] = int(info.scan_engine), int(info.adaptive)
            ms, c0 = timed(h, text, args.reps)
            ad = h.adapt()
            rec["chosen"] = dict(first_ms=round(ms[0], 4), per_launch_ms=[round(x, 3) for x in ms], last_ms=round(min(ms[-3:]), 4), matches=c0,
                                 engine_now=int(ad.engine), flips=int(ad.flips), reports=int(ad.reports),
                                 ms_per_gib=[round(x, 4) for x in ad.ms_per_gib], events_per_4k=[round(x, 3) for x in ad.events_per_4k],
                                 est=[round(x, 4) for x in ad.est_ms_per_gib], verify_density_x4096=round(ad.verify_density * 4096, 3))
            for eng, label in ((S.ALGO_AC, "automaton"), (S.ALGO_WM, "filter"), (S.ENGINE_AC_FLAT, "flat_automaton")):
                try:
                    h.set_scan_engine(eng)
                except S.SmhError as e:
                    rec[label] = "n/a: " + str(e)[:60]
                    continue
                ms, c = timed(h, text, 4)
                rec[label] = dict(ms=round(min(ms[1:]), 4), matches=c, equal=c == c0)
            h.close()
            print(json.dumps(rec), flush=True)
