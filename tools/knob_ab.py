"""Development aid: one configs[4]-shaped set under several SMH_WM_TUNE settings read at LAUNCH time (same handle, launches interleaved).
usage: knob_ab.py M "knobA" "knobB" ...   ("" = default)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch  # noqa: E402
import smatcher_hip as S  # noqa: E402
m = int(sys.argv[1])
knobs = sys.argv[2:] or [""]
n = 4 << 30
dev = torch.device("cuda", 0)
text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
S.corpus_text_device(text.data_ptr(), n, 42, 256)
torch.cuda.synchronize()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
pat = S.corpus_patterns(m, 100000, 7, 256, 42, n, 2)
os.environ["SMH_ADAPT"] = "0"
h = S.WmTables.from_patterns(pat, m, 100000, 256)
ts = {k: [] for k in knobs}
counts = {}
for it in range(14):
    for k in knobs:
        os.environ["SMH_WM_TUNE"] = k
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        h.scan_device(text.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st)
        b.record()
        torch.cuda.synchronize()
        if it >= 4:
            ts[k].append(a.elapsed_time(b))
        counts[k] = int(cnt.item())
base = sorted(ts[knobs[0]])[len(ts[knobs[0]]) // 2]
for k in knobs:
    med = sorted(ts[k])[len(ts[k]) // 2]
    print("m=%d knob=%-12r median %.4f ms  (%.3f of %r)  count %d" % (m, k, med, med / base, knobs[0], counts[k]), flush=True)
