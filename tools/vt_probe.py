"""Development aid: configs[4] (100 000 byte patterns, 4 GiB) with the verify table capped at 1 MiB (rounds 2-3: SMH_WM_TUNE=vt=1m) and at its default size."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch  # noqa: E402
import smatcher_hip as S  # noqa: E402
n = 4 << 30
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.corpus_text_device(text.data_ptr(), n, 42, 256)
torch.cuda.synchronize()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for m in (5, 8, 12, 20):
    pat = S.corpus_patterns(m, 100000, 7, 256, 42, n, 2)
    hs = {}
    for tune in ("vt=1m", ""):
        os.environ["SMH_WM_TUNE"] = tune
        hs[tune] = S.WmTables.from_patterns(pat, m, 100000, 256)
    os.environ["SMH_WM_TUNE"] = ""
    ts = {k: [] for k in hs}
    counts = {}
    for it in range(14):
        for k, h in hs.items():
            cnt.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
            b.record()
            torch.cuda.synchronize()
            if it >= 4:
                ts[k].append(a.elapsed_time(b))
            counts[k] = int(cnt.item())
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    print("m=%d  1 MiB cap %.4f ms (verify slots %d)  default %.4f ms (slots %d)  ratio %.3f  counts equal %s" % (
        m, med["vt=1m"], hs["vt=1m"].info().verify_slots, med[""], hs[""].info().verify_slots, med[""] / med["vt=1m"], counts[""] == counts["vt=1m"]), flush=True)
    for h in hs.values():
        h.close()
