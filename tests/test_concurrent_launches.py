"""Launches of ONE handle from two host threads on two streams: the adaptive state (csrc/smh_adapt.h) is
guarded by a per-handle-and-device mutex, every reporting launch has its own ticket slot and nonce (csrc/smh_stats.h), and the
library orders the handle's launches on the device across streams (smh_runtime.hip adapt_order_before).  Counts must be those of
a single-threaded run, and on uniform text the engine must stay where the compile put it.  The host state machine itself runs
under ThreadSanitizer on the CPU: tools/tsan_adapt.sh (test below)."""
import os
import subprocess
import threading

import numpy as np
import pytest
import torch

import emu_lib  # noqa: F401  (sys.path)
import oracle_lib as O
import smatcher_hip as S
from perf import perf_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adaptive_state_machine_is_tsan_clean_on_the_cpu():
    r = subprocess.run([os.path.join(ROOT, "tools", "tsan_adapt.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tsan: clean with the mutex" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.perf
@pytest.mark.parametrize("entry", ["ac", "wm"])
def test_two_threads_two_streams_one_handle(entry):
    n, m, p, sigma, launches = 256 << 20, 16, 1000, 4, 50  # (launches long enough for their rate to mean something: tests/test_adaptive.py uses the same size)
    dev = torch.device("cuda", 0)
    texts = {}
    for name, kind in (("uniform", S.CORPUS_UNIFORM), ("repeats", S.CORPUS_DNA_REPEATS)):
        t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        S.corpus_text_device(t.data_ptr(), n, 42, sigma, 0, kind, torch.cuda.current_stream().cuda_stream)
        texts[name] = t
    torch.cuda.synchronize()
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2, S.CORPUS_DNA_REPEATS)
    want = {k: O.oracle_ac(pat, m, p, sigma, v[:n].cpu().numpy())[0] for k, v in texts.items()}
    make = S.AcAutomaton if entry == "ac" else S.WmTables
    for name in ("uniform", "repeats"):
        h = make.from_patterns(pat, m, p, sigma)
        assert h.info().adaptive == 1
        first = h.info().scan_engine
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        counts = [torch.zeros(launches, dtype=torch.int64, device=dev) for _ in range(2)]
        errors = []

        def work(i):
            try:
                torch.cuda.set_device(0)
                for k in range(launches):
                    h.scan_device(texts[name].data_ptr(), n, counts[i].data_ptr() + 8 * k, S.VARIANT_TUNED, streams[i].cuda_stream)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for i in range(2):
            got = counts[i].cpu().numpy()
            assert np.all(got == want[name]), (name, i, np.unique(got), want[name])
        ad = h.adapt()
        assert ad.reports >= 2
        if name == "uniform":  # a property of measured rates: recorded in a parity run, enforced under -m "gpu and perf" (tests/perf.py)
            perf_check(ad.flips == 0 and ad.engine == first, "uniform text, two threads: %d engine flips, engine %d (compiled: %d)" % (ad.flips, ad.engine, first))
        else:
            assert ad.engine in (S.ENGINE_AC_FLAT, S.ENGINE_KEYS), ad.engine  # the text-independent engine, as single-threaded
        # a forced depth-cut automaton plan shares ONE candidate queue between its launches: ordered on the device, the counts hold
        if entry == "ac":
            h.set_scan_engine(S.ALGO_AC)
            for c in counts:
                c.zero_()
            threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            torch.cuda.synchronize()
            assert not errors, errors
            for i in range(2):
                assert np.all(counts[i].cpu().numpy() == want[name]), (name, "automaton kernels", i)
        h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sigma,m,p", [(256, 12, 10000), (4, 16, 8000)])
def test_scans_and_positions_of_one_handle_on_two_streams(sigma, m, p):
    """Round 6: smh_wm_positions shares the handle's per-device survivor queue with smh_wm_scan, so it takes the
    same mutex and the same ordering across streams.  One thread scans, the other asks for positions, same handle, filter
    kernels forced (the queue's users), 40 launches each: every count and every position list equals the single-stream one."""
    n, launches = 16 << 20, 40
    dev = torch.device("cuda", 0)
    t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    S.corpus_text_device(t.data_ptr(), n, 42, sigma, 0, S.CORPUS_UNIFORM, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pat = S.corpus_patterns(m, p, 7, sigma, 42, n, 2)
    h = S.WmTables.from_patterns(pat, m, p, sigma)
    h.set_scan_engine(S.ALGO_WM)
    one = torch.zeros(1, dtype=torch.int64, device=dev)
    h.scan_device(t.data_ptr(), n, one.data_ptr(), S.VARIANT_TUNED, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = int(one.item())
    assert want == O.oracle_wu(pat, m, p, sigma, t[:n].cpu().numpy())[0] and want > 100
    ref = torch.zeros(want + 8, dtype=torch.int64, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    h.positions_device(t.data_ptr(), n, ref.data_ptr(), want + 8, cur.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(cur.item()) == want
    ref_sorted = np.sort(ref[:want].cpu().numpy())
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    counts = torch.zeros(launches, dtype=torch.int64, device=dev)
    cursors = torch.zeros(launches, dtype=torch.int64, device=dev)
    outs = torch.zeros((launches, want + 8), dtype=torch.int64, device=dev)
    errors = []

    def scans():
        try:
            torch.cuda.set_device(0)
            for k in range(launches):
                h.scan_device(t.data_ptr(), n, counts.data_ptr() + 8 * k, S.VARIANT_TUNED, streams[0].cuda_stream)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def positions():
        try:
            torch.cuda.set_device(0)
            for k in range(launches):
                h.positions_device(t.data_ptr(), n, outs[k].data_ptr(), want + 8, cursors.data_ptr() + 8 * k, streams[1].cuda_stream)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=scans), threading.Thread(target=positions)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert np.all(counts.cpu().numpy() == want), np.unique(counts.cpu().numpy())
    assert np.all(cursors.cpu().numpy() == want), np.unique(cursors.cpu().numpy())
    got = outs.cpu().numpy()
    for k in range(launches):
        assert np.array_equal(np.sort(got[k, :want]), ref_sorted), k
    h.close()

