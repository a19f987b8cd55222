"""A tuned scan is a plain kernel launch on the caller's stream once the handle's tables are resident on the device: it can be
captured into a HIP graph and replayed (the launch-bound end of the spectrum: many short texts, one graph launch per batch)."""
import os
import sys

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["ac", "wm"])
def test_scans_replay_from_a_captured_graph(entry):
    import torch
    n, m, p, sigma = 48 << 20, 16, 1000, 4
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    t = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    S.corpus_text_device(t.data_ptr(), n, 42, sigma)
    torch.cuda.synchronize()
    host = t[:n].cpu().numpy()
    want, want_half = O.oracle_ac(pat, m, p, sigma, host)[0], O.oracle_ac(pat, m, p, sigma, host[:n // 2])[0]
    h = (S.AcAutomaton if entry == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):  # first scans outside the capture: the table set goes up with blocking copies
        h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st.cuda_stream)
    torch.cuda.synchronize()
    assert int(cnt[0].item()) == want
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        cnt.zero_()
        h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, st.cuda_stream)
        h.scan_device(t.data_ptr(), n // 2, cnt.data_ptr() + 8, S.VARIANT_TUNED, st.cuda_stream)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert [int(x) for x in cnt.tolist()] == [want, want_half]
    h.close()
