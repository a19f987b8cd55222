"""Development aid: the byte-gram kernels with and without survivors -- 100 000 patterns over bytes 128..255 against a text
of bytes 0..127 (no gram of the text is in the set: the bare filter scan) and against the usual 256-symbol text."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import torch, numpy as np, smatcher_hip as S
dev = torch.device("cuda", 0); n = 4 << 30
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for sigma_text in (128, 256):
    text = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    S.corpus_text_device(text.data_ptr(), n, 42, sigma_text); torch.cuda.synchronize()
    for m in (5, 8, 12, 20):
        pat = S.corpus_patterns(m, 100000, 9, 256, 42, n, 2)
        if sigma_text == 128: pat = pat | 0x80
        wm = S.WmTables.from_patterns(pat, m, 100000, 256)
        ts = []
        for it in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cnt.zero_(); a.record(); wm.scan_device(text.data_ptr(), n, cnt.data_ptr(), 0, st); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ad = wm.adapt()
        print("text alphabet %d m=%d: %.4f ms per 4 GiB (%.3f of peak) survivors/4KiB %.2f matches %d kind %d" % (sigma_text, m, sorted(ts)[3], n / sorted(ts)[3] / 1e-3 / 8e12, ad.events_per_4k[1], int(cnt.item()), wm.info().gram_kind), flush=True)
        wm.close()
