"""smh_multi_*: one process driving several GPUs -- byte-range shards resident per device, kernels side by side,
ONE RCCL all-reduce of the 64-bit counts (the reference's MPI_Scatterv / MPI_Reduce, main.c:464-489, 654-657).

CPU box: the entry points exist and fail cleanly without a device.  GPU box: with one device the whole path runs
with an RCCL communicator of one rank; with two or more devices visible the shards really spread (skipped at one)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
import smatcher_hip as S  # noqa: E402


def test_multi_entry_points_exist_and_fail_cleanly_without_devices():
    for name in ("smh_multi_create", "smh_multi_device_count", "smh_multi_uses_rccl", "smh_multi_load_text",
                 "smh_multi_generate_text", "smh_multi_ac_count", "smh_multi_wm_count", "smh_multi_free"):
        getattr(S.lib, name)
    if S.device_count() == 0:
        with pytest.raises(S.SmhError, match="visible"):
            S.MultiGpu(1)
    with pytest.raises(S.SmhError):
        S.MultiGpu(0)
    with pytest.raises(S.SmhError):
        S.MultiGpu(2, devices=[0, 0])  # a device twice / not visible: refused either way


def _case(n, m, p, sigma):
    text = S.corpus_text(n, 42, sigma)
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    want, _ = O.oracle_ac(pat, m, p, sigma, text)
    return text, pat, want


@pytest.mark.gpu
@pytest.mark.parametrize("n_devices", [1, 2, 4, 8])
def test_multi_device_counts_match_the_oracle(n_devices):
    if S.device_count() < n_devices:
        pytest.skip("%d device(s) visible" % S.device_count())
    n, m, p, sigma = 6_000_007, 16, 500, 4
    text, pat, want = _case(n, m, p, sigma)
    mg = S.MultiGpu(n_devices)
    assert mg.devices == n_devices and mg.uses_rccl
    mg.load_text(text, 63)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    for handle, fn in ((ac, mg.ac_count), (wm, mg.wm_count)):
        total, per, secs = fn(handle)
        assert total == want and sum(per) == want and len(per) == n_devices and secs > 0
        # every device's count is the oracle's count of its byte range (main.c:467-477)
        for r in range(n_devices):
            b, e = S.shard_range(n, n_devices, r, m)
            assert per[r] == O.oracle_ac(pat, m, p, sigma, text[b:e])[0]
    # a second, shorter pattern set over the SAME resident shards; a longer one than the halo is refused
    pat8 = S.corpus_patterns(8, 100, 9, sigma, 42, n, 2)
    ac8 = S.AcAutomaton.from_patterns(pat8, 8, 100, sigma)
    assert mg.ac_count(ac8)[0] == O.oracle_ac(pat8, 8, 100, sigma, text)[0]
    pat65 = S.corpus_patterns(65, 10, 9, sigma, 42, n, 2)
    with pytest.raises(S.SmhError, match="halo"):
        mg.ac_count(S.AcAutomaton.from_patterns(pat65, 65, 10, sigma))
    # the synthetic corpus generated shard by shard on the devices equals the host corpus
    mg.generate_text(n, 42, sigma, 31)
    assert mg.ac_count(ac)[0] == want
    # host-side sum instead of the communicator: same numbers
    mg2 = S.MultiGpu(n_devices, flags=S.MULTI_NO_RCCL)
    assert not mg2.uses_rccl
    mg2.load_text(text, 15)
    assert mg2.ac_count(ac)[0] == want
    mg2.close()
    mg.close()
