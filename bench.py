#!/usr/bin/env python3
"""bench.py -- Gbit/s of text scanned by the MI355X multi-pattern matcher.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the driver may launch it
through torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE in the environment); from a
plain shell with WORLD_SIZE unset, `--gpus N` fans out by itself: the parent -- which never touches the
GPU -- starts N fresh child processes, one rank each, over RCCL (backend "nccl"), and exits non-zero if any
child fails.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): 1 GiB of synthetic
4-letter DNA text per GPU, resident in HBM before the timed region, 1 000 patterns per set with
pattern lengths 8-32.  The reference API carries ONE pattern length per run (smatcher.h:89-106),
so "len 8-32" is a sweep of fixed-length sets, m = 8, 16, 32; one STEP = one
Aho-Corasick pass over the rank's text for each of the three sets (3 GiB of text scanned per GPU
per step), and for N > 1 one RCCL all-reduce of the three 64-bit counts (the reference's
MPI_Reduce, main.c:656).  value = bits scanned by all ranks / wall time of the K steps.

N > 1 is weak scaling: every rank holds its own 1 GiB byte range (+ m-1 halo) of one N GiB text
(shard formula main.c:467-477); there is no data-path collective.

Extra objects on the JSON line:
  roofline               HBM bound; achieved = algorithmic bytes per launch (1 byte per text symbol) / mean
                         launch duration from events on the launch stream; traffic from the committed
                         rocprofv3 --pmc passes when they were taken on THIS build of the kernels
  cpu_baseline           the reference's own compiled search_ac (oracle/_ref; the oracle port when absent),
                         one thread, on a bounded prefix of the same text
  cpu_baseline_wm        the same for search_wu2 (wu/wu.c:151-209) beside the WM configuration
  cpu_baseline_all_cores both, fanned out over the host's cores by byte range (main.c:467-477)
  ac / wm                per-configuration rates (WM = BASELINE configs[2]: same text, 10 000 x m=8)
  ac_8000_patterns       BASELINE configs[3] shape on one GPU (a 4 GiB shard, 8 000 patterns, m = 8/16/32)
  wm_ascii               BASELINE configs[4] shape on one GPU (a 4 GiB shard of 256-symbol text, 100 000
                         patterns, m = 5/12/20)
  verified               EVERY `matches` above against a CPU count of the same text in the same run: the
                         restated search_ac / search_wu2 (oracle/, pinned to the reference) over byte-range
                         shards on all host cores
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd")
sys.path.insert(0, PKG)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TEXT_SEED, PAT_SEED, SIGMA = 42, 7, 4
AC_LENGTHS = (8, 16, 32)
AC_PATTERNS = 1000
WM_PATTERNS, WM_LENGTH = 10000, 8
C4_PATTERNS = 8000
C5_PATTERNS, C5_LENGTHS, C5_SIGMA = 100000, (5, 12, 20), 256


def kernel_build_id():
    """Digest of the kernel and host sources the library was built from: the traffic figures in
    profiles/hbm_traffic.json are only quoted when they were measured on the same sources."""
    h = hashlib.sha256()
    src = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".inc", ".hip", ".c")):
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:12]


def measured_traffic(info):
    """-> (HBM bytes per launch or None, source string) for the AC kernel instance `info` (smh_ac_info) selects.
    bench.py cannot read PMC counters itself; profiles/hbm_traffic.json is produced from this same command under
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/collect_counters.sh; FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md) and carries the build id it was taken on."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no committed counter pass"
    rec = json.load(open(path))
    have, want = rec.get("build_id"), kernel_build_id()
    if have != want:
        return None, "profiles/hbm_traffic.json was taken on build %s, this is build %s: not quoted" % (have, want)
    halo = info.scan_depth - 1
    hc = 1 if halo <= 16 else (2 if halo <= 32 else 4)
    entry = "unsigned short" if (info.scan_stride == 2 or info.lds_rows <= 32768) else "unsigned int"
    stride = 3 if info.scan_full_rows else info.scan_stride  # template value of the hybrid image
    prefix = "ac_dfa_kernel<%s, 4, %d, %d, %s," % (entry, stride, hc, "true" if info.scan_exact else "false")
    for name, k in rec.get("kernels", {}).items():
        if name.startswith(prefix) and not name.endswith("true>"):  # "..., true>" = the positions-mode instance
            return k["hbm_bytes"], "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, build %s, %s)" % (
                have, rec.get("profile", "?"))
    return None, "no counter pass for " + prefix


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---------------------------------------------------------------------------------------------------------
# CPU side: the checker.  The only place bench.py touches oracle/ (through tests/oracle_lib.py).
class Cpu:
    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from concurrent.futures import ThreadPoolExecutor
        self.O = O
        self.cores = max(1, min(len(os.sched_getaffinity(0)), 64))
        self.pool = ThreadPoolExecutor(self.cores)
        self.kind = "reference" if O.have_ref() else "port"
        self.model = cpu_model()

    # --- timed baselines (reference when present) ---
    def ac_serial(self, pats, p, sigma, text):
        """search_ac (ac/ac.c:198-222), one thread, every set in `pats` over `text`."""
        O, secs, counts = self.O, 0.0, {}
        for m, pat in pats.items():
            if self.kind == "reference":
                cnt, _, _, ts = O.ref_ac(pat, m, p, sigma, text)
            else:
                _, tabs = O.oracle_ac(pat, m, p, sigma)
                t0 = time.perf_counter()
                cnt = O.oracle_ac_search_tables(text, sigma, tabs)
                ts = time.perf_counter() - t0
            secs += ts
            counts[m] = cnt
        return secs, counts

    def wm_serial(self, pat, m, p, sigma, text):
        """search_wu2 (wu/wu.c:151-209), one thread."""
        O = self.O
        if self.kind == "reference":
            cnt, _, _, ts = O.ref_wu(pat, m, p, sigma, text, flat=True)
        else:
            csr = O.WMTablesCSR(pat, m, p, sigma)
            t0 = time.perf_counter()
            cnt = csr.search(text)
            ts = time.perf_counter() - t0
        return ts, cnt

    def ac_all_cores_reference(self, pats, p, sigma, text, want):
        """The reference search fanned out by byte range with an m-1 halo -- its own MPI decomposition
        (main.c:467-477) with threads for ranks; time = the slowest shard's search_ac per set, summed."""
        O, n, ok, secs, wall0 = self.O, len(text), True, 0.0, time.perf_counter()
        for m, pat in pats.items():
            ranges = [O.shard_range(n, self.cores, r, m) for r in range(self.cores)]
            parts = list(self.pool.map(lambda be: O.ref_ac(pat, m, p, sigma, text[be[0]:be[1]]), ranges))
            ok = ok and sum(q[0] for q in parts) == want[m]
            secs += max(q[3] for q in parts)
        return secs, ok, time.perf_counter() - wall0

    def wm_all_cores_reference(self, pat, m, p, sigma, text, want):
        O, n = self.O, len(text)
        ranges = [O.shard_range(n, self.cores, r, m) for r in range(self.cores)]
        parts = list(self.pool.map(lambda be: O.ref_wu(pat, m, p, sigma, text[be[0]:be[1]], flat=True), ranges))
        return max(q[3] for q in parts), sum(q[0] for q in parts) == want

    # --- full-text verification (restated search, tables built once and shared by the threads) ---
    def ac_count(self, pat, m, p, sigma, text):
        O, n = self.O, len(text)
        _, tabs = O.oracle_ac(pat, m, p, sigma)
        pieces = max(self.cores * 4, 1)
        ranges = [O.shard_range(n, pieces, r, m) for r in range(pieces)]
        return sum(self.pool.map(lambda be: O.oracle_ac_search_tables(text[be[0]:be[1]], sigma, tabs), ranges))

    def wm_count(self, pat, m, p, sigma, text):
        O, n = self.O, len(text)
        csr = O.WMTablesCSR(pat, m, p, sigma)
        pieces = max(self.cores * 4, 1)
        ranges = [O.shard_range(n, pieces, r, m) for r in range(pieces)]
        return sum(self.pool.map(lambda be: csr.search(text[be[0]:be[1]]), ranges))


def spawn_ranks(n):
    """WORLD_SIZE unset and --gpus N > 1: start N fresh rank processes (this process has not touched the GPU and
    never does), pass rank 0's stdout through, fail if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, deadline = 0, time.time() + 3600
    live = list(procs)
    while live and time.time() < deadline:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:  # a failed rank would leave the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    for p in live:
        p.kill()
        rc = rc or 1
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib-per-gpu", type=int, default=1024, help="text bytes per GPU in MiB (BASELINE: 1024)")
    ap.add_argument("--cpu-sample-mib", type=int, default=96, help="prefix the serial search_ac baseline runs on")
    ap.add_argument("--cpu-wm-sample-mib", type=int, default=16, help="prefix the serial search_wu2 baseline runs on")
    ap.add_argument("--shard-mib", type=int, default=4096, help="per-GPU shard of the 32 GB configurations (configs[3], [4])")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines and the full-text verification")
    ap.add_argument("--no-wm", action="store_true", help="skip the side configurations (WM, configs[3], configs[4])")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return

    import numpy as np
    import torch
    import torch.distributed as dist
    import smatcher_hip as S
    import sharded

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available() or S.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the scan path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    per_gpu = args.mib_per_gpu << 20
    n_total = per_gpu * world
    stream = torch.cuda.current_stream().cuda_stream
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def corpus(n, offset, sigma):
        t = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        rc = S.lib.smh_corpus_text_device(C.c_void_p(t.data_ptr()), n, offset, TEXT_SEED, sigma, C.c_void_p(stream))
        if rc != 0:
            raise SystemExit("corpus generation failed: " + S.lib.smh_last_error().decode())
        return t

    def timed(launch, reps, counter):
        """`reps` launches bracketed by events on the launch stream -> list of ms"""
        launch()
        torch.cuda.synchronize()
        evs = [(ev(), ev()) for _ in range(reps)]
        for a, b in evs:
            counter.zero_()
            a.record()
            launch()
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    def rate(nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return dict(GBps=round(gbs, 1), Gbit_s=round(8 * gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4))

    # ---- pattern sets (host) and compiled automata
    pats = {m: S.corpus_patterns(m, AC_PATTERNS, PAT_SEED, SIGMA, TEXT_SEED, n_total, 2) for m in AC_LENGTHS}
    acs = {m: S.AcAutomaton.from_patterns(pats[m], m, AC_PATTERNS, SIGMA) for m in AC_LENGTHS}
    halo = max(AC_LENGTHS) - 1

    # ---- this rank's byte range of the N GiB text, generated in HBM (never crosses PCIe)
    begin = rank * per_gpu
    shard_ends = {m: sharded.shard_for_rank(n_total, world, rank, m) for m in AC_LENGTHS}
    for m in AC_LENGTHS:
        assert shard_ends[m][0] == begin
    n_alloc = min(per_gpu + halo, n_total - begin)
    text = corpus(n_alloc, begin, SIGMA)
    counts = torch.zeros(len(AC_LENGTHS), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def shard_len(m):
        b, e = shard_ends[m]
        return e - b

    # every step has its own count buffer: its all-reduce is started behind its three scans and runs on RCCL's stream
    # while the next step's scans run on ours; all of them are waited for inside the timed region
    step_counts = torch.zeros((args.warmup + args.steps + 1, len(AC_LENGTHS)), dtype=torch.int64, device=dev)

    def step(k, events=None):
        c = step_counts[k]  # zero since its allocation: every step accumulates into a row of its own
        if events is not None:
            events[0].record()  # one event between consecutive launches: the end of one is the start of the next
        for i, m in enumerate(AC_LENGTHS):
            acs[m].scan_device(text.data_ptr(), shard_len(m), c.data_ptr() + 8 * i, S.VARIANT_TUNED, stream)
            if events is not None:
                events[i + 1].record()
        return sharded.reduce_count_async(c)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sharded.finish([step(k) for k in range(args.warmup)])
    evs = [[ev() for _ in range(len(AC_LENGTHS) + 1)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    sharded.finish([step(args.warmup + k, evs[k]) for k in range(args.steps)])
    barrier()
    elapsed = time.perf_counter() - t0
    counts = step_counts[args.warmup + args.steps - 1] if args.steps else step_counts[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_counts = [int(x) for x in counts.tolist()]
    # per-GPU counts for the report: one untimed pass without the reduce, then one small all-gather
    counts = step_counts[-1]
    counts.zero_()
    for i, m in enumerate(AC_LENGTHS):
        acs[m].scan_device(text.data_ptr(), shard_len(m), counts.data_ptr() + 8 * i, S.VARIANT_TUNED, stream)
    torch.cuda.synchronize()
    per_gpu_counts = sharded.gather_counts(counts).tolist()
    local_counts = [int(x) for x in counts.tolist()]

    # per-launch durations (ms) from the events on the launch stream
    kern_ms = {m: [evs[k][i].elapsed_time(evs[k][i + 1]) for k in range(args.steps)] for i, m in enumerate(AC_LENGTHS)}
    bits_per_step = 8.0 * sum(shard_len(m) for m in AC_LENGTHS)
    if world > 1:
        t = torch.tensor([bits_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bits_per_step = float(t.item())
    value = bits_per_step * args.steps / elapsed / 1e9

    out = None
    verify = []  # (name, algorithm, patterns, m, p, sigma, device text tensor, n, gpu count): checked at the end
    if rank == 0:
        mean = lambda xs: sum(xs) / len(xs)
        ac_detail = {}
        for i, m in enumerate(AC_LENGTHS):
            info = acs[m].info()
            ms = mean(kern_ms[m])
            ac_detail["m%d" % m] = dict(kernel_ms=round(ms, 4), median_ms=round(sorted(kern_ms[m])[len(kern_ms[m]) // 2], 4),
                                        min_ms=round(min(kern_ms[m]), 4), **rate(shard_len(m), ms),
                                        dfa_rows=info.rows, lds_rows=info.lds_rows, lds_bytes=info.lds_bytes,
                                        scan_stride=info.scan_stride, scan_depth=info.scan_depth,
                                        scan_exact=info.scan_exact, scan_full_rows=info.scan_full_rows,
                                        scan_engine="suffix-filter kernels" if info.scan_engine == S.ALGO_WM else "automaton kernels",
                                        matches=total_counts[i])
            verify.append(("ac.m%d" % m, "ac", pats[m], m, AC_PATTERNS, SIGMA, text, shard_len(m), local_counts[i]))
        dom = max(AC_LENGTHS, key=lambda m: mean(kern_ms[m]))
        dom_ms = mean(kern_ms[dom])
        achieved = shard_len(dom) / (dom_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(acs[dom].info())
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                        kernel="ac_dfa_kernel (m=%d set)" % dom, launch_ms=round(dom_ms, 4),
                        algorithmic_bytes_per_launch=shard_len(dom))
        out = {
            "metric": "Gbit/s text scanned (AC and WM) at 1/2/4/8 MI355X; % HBM roofline",
            "value": round(value, 2), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "AC on MI355X: %d MiB synthetic DNA text per GPU resident in HBM, %d patterns per set, "
                                   "pattern lengths 8-32 as fixed-length sets m=8/16/32 (BASELINE configs[1]); "
                                   "step = 3 scans + count all-reduce" % (args.mib_per_gpu, AC_PATTERNS),
                       "text_bytes_per_gpu": per_gpu, "alphabet": SIGMA, "patterns": AC_PATTERNS,
                       "pattern_lengths": list(AC_LENGTHS), "text_seed": TEXT_SEED, "pattern_seed": PAT_SEED,
                       "sharding": "byte-range x%d, m-1 halo, RCCL sum of counts" % world},
            "roofline": roofline, "ac": ac_detail, "device": S.device_name(), "kernel_build_id": kernel_build_id(),
            "per_gpu_matches": {"m%d" % m: [int(r[i]) for r in per_gpu_counts] for i, m in enumerate(AC_LENGTHS)},
        }

    # ---- what a pure streaming read of the same 1 GiB reaches on this device, same run
    if rank == 0:
        probe = torch.zeros(1, dtype=torch.int64, device=dev)
        pms = sorted(timed(lambda: S.lib.smh_stream_read_probe(C.c_void_p(text.data_ptr()), per_gpu, C.c_void_p(probe.data_ptr()),
                                                               C.c_void_p(stream)), 6, probe))[2]
        out["stream_read"] = dict(kernel="smh_stream_read_kernel (16-byte loads, XOR, no table work)", ms=round(pms, 4),
                                  **{k: v for k, v in rate(per_gpu, pms).items() if k != "Gbit_s"})
        out["roofline"]["of_stream_read"] = round(out["roofline"]["achieved"] / out["stream_read"]["GBps"], 4)

    # ---- match positions: the m=16 set's END columns into a device buffer, same text
    if rank == 0:
        m_pos = 16
        cap = max(1024, 2 * int(local_counts[AC_LENGTHS.index(m_pos)]))
        pbuf = torch.zeros(cap, dtype=torch.int64, device=dev)
        pcur = torch.zeros(1, dtype=torch.int64, device=dev)
        pms = sorted(timed(lambda: acs[m_pos].positions_device(text.data_ptr(), shard_len(m_pos), pbuf.data_ptr(), cap,
                                                               pcur.data_ptr(), stream), 6, pcur))[2]
        out["positions"] = dict(workload="smh_ac_positions, m=%d set, same text: END columns of all matches" % m_pos,
                                kernel_ms=round(pms, 4), GBps=rate(shard_len(m_pos), pms)["GBps"], matches=int(pcur.item()),
                                equals_count=int(pcur.item()) == local_counts[AC_LENGTHS.index(m_pos)])

    # ---- WM side measurement (BASELINE configs[2]: same text, 10 000 patterns of length 8)
    wpat = None
    if not args.no_wm:
        wpat = S.corpus_patterns(WM_LENGTH, WM_PATTERNS, PAT_SEED + 1, SIGMA, TEXT_SEED, n_total, 2)
        wm = S.WmTables.from_patterns(wpat, WM_LENGTH, WM_PATTERNS, SIGMA)
        wb, we = sharded.shard_for_rank(n_total, world, rank, WM_LENGTH)
        wcount = torch.zeros(1, dtype=torch.int64, device=dev)
        wms = timed(lambda: wm.scan_device(text.data_ptr(), we - wb, wcount.data_ptr(), S.VARIANT_TUNED, stream), args.steps, wcount)
        wlocal = int(wcount.item())
        sharded.reduce_count(wcount)
        if rank == 0:
            wi = wm.info()
            ms = sum(wms) / len(wms)
            out["wm"] = dict(workload="WM: same text, %d patterns of length %d (BASELINE configs[2]); per-GPU kernel rate"
                                      % (WM_PATTERNS, WM_LENGTH),
                             kernel_ms=round(ms, 4), min_ms=round(min(wms), 4), **rate(we - wb, ms), matches=int(wcount.item()),
                             block_symbols=wi.block_symbols, filter_log2=wi.filter_log2, filter_exact=wi.filter_exact,
                             shift_zero="%d/%d" % (wi.shift_zero, wi.shiftsize))
            verify.append(("wm", "wm", wpat, WM_LENGTH, WM_PATTERNS, SIGMA, text, we - wb, wlocal))
        # the headline's longer pattern sets (m = 16, 32; the same 1000 patterns) through the Wu-Manber entry point
        if rank == 0 and world == 1:
            wl = {}
            for m in AC_LENGTHS[1:]:
                wml = S.WmTables.from_patterns(pats[m], m, AC_PATTERNS, SIGMA)
                wcount.zero_()
                mls = timed(lambda: wml.scan_device(text.data_ptr(), shard_len(m), wcount.data_ptr(), S.VARIANT_TUNED, stream), args.steps, wcount)
                li = wml.info()
                ms = sum(mls) / len(mls)
                wl["m%d" % m] = dict(kernel_ms=round(ms, 4), min_ms=round(min(mls), 4), **rate(shard_len(m), ms), matches=int(wcount.item()),
                                     scan_engine="automaton kernels" if li.scan_engine == S.ALGO_AC else "suffix-filter kernels",
                                     gram_planes=li.gram_planes)
                verify.append(("wm_long.m%d" % m, "wm", pats[m], m, AC_PATTERNS, SIGMA, text, shard_len(m), int(wcount.item())))
                del wml
            out["wm_long"] = dict(workload="WM: same text, the headline's %d-pattern sets of length %s through the Wu-Manber entry "
                                           "point (q-gram shift-or filter in LDS + staged verify)" % (AC_PATTERNS, "/".join(str(m) for m in AC_LENGTHS[1:])), **wl)

    # ---- BASELINE configs[1] read literally: ONE set of 1000 patterns whose lengths run from 8 to 32 (40 per length),
    #      through the pattern-set entry points (smh_pset_*: the reference API carries one length per run)
    mixed = None
    if not args.no_wm and rank == 0 and world == 1:
        mlens, mpats = [], []
        for L in range(8, 33):
            mpats.append(S.corpus_patterns(L, 40, PAT_SEED + 100 + L, SIGMA, TEXT_SEED, n_total, 2))
            mlens += [L] * 40
        mixed = (np.concatenate(mpats), np.array(mlens, dtype=np.uint32))
        mcount = torch.zeros(1, dtype=torch.int64, device=dev)
        mobj = {}
        for name, algo in (("ac", S.ALGO_AC), ("wm", S.ALGO_WM)):
            ps = S.PatternSet(mixed[0], mixed[1], SIGMA, algo)
            mls = timed(lambda: ps.scan_device(text.data_ptr(), per_gpu, mcount.data_ptr(), stream), args.steps, mcount)
            ms = sum(mls) / len(mls)
            mobj[name] = dict(kernel_ms=round(ms, 4), min_ms=round(min(mls), 4), **rate(per_gpu, ms), matches=int(mcount.item()),
                              one_pass=int(ps.info().one_pass), classes=int(ps.info().classes))
            ps.close()
        out["mixed_8_32"] = dict(workload="BASELINE configs[1] read as ONE set: 1000 patterns, 40 of each length 8..32, same text, "
                                          "scanned in one pass; count = sum over length classes of the reference's count", **mobj)

    # ---- the 32 GB configurations, one GPU's shard of each (rank 0 of a single-GPU run only)
    side = not args.no_wm and world == 1
    shard = args.shard_mib << 20
    if side:
        # BASELINE configs[3]: AC, 8000 patterns; 32 GB over 8 GPUs = a 4 GiB byte range per GPU
        text4 = text if shard == per_gpu else corpus(shard, 0, SIGMA)
        c4, c4cnt = {}, torch.zeros(1, dtype=torch.int64, device=dev)
        for m4 in AC_LENGTHS:
            p4 = S.corpus_patterns(m4, C4_PATTERNS, PAT_SEED + 3, SIGMA, TEXT_SEED, shard, 2)
            ac4 = S.AcAutomaton.from_patterns(p4, m4, C4_PATTERNS, SIGMA)
            ms4 = sorted(timed(lambda: ac4.scan_device(text4.data_ptr(), shard, c4cnt.data_ptr(), S.VARIANT_TUNED, stream), 5, c4cnt))[2]
            i4 = ac4.info()
            c4["m%d" % m4] = dict(kernel_ms=round(ms4, 4), **rate(shard, ms4), matches=int(c4cnt.item()),
                                  scan_engine="suffix-filter kernels" if i4.scan_engine == S.ALGO_WM else "automaton kernels",
                                  scan_stride=i4.scan_stride, scan_depth=i4.scan_depth)
            verify.append(("ac_8000_patterns.m%d" % m4, "ac", p4, m4, C4_PATTERNS, SIGMA, text4, shard, int(c4cnt.item())))
            del ac4
        out["ac_8000_patterns"] = dict(workload="AC: %d MiB of DNA text (one GPU's byte range of BASELINE configs[3]: 32 GB over 8 "
                                                "GPUs), 8000 patterns per set, m=8/16/32; scan_engine says which kernels served "
                                                "the Aho-Corasick entry point" % args.shard_mib, **c4)
        # BASELINE configs[4]: WM, 256-symbol alphabet, 100 000 patterns, lengths 5-20 as fixed-length sets
        text5 = corpus(shard, 0, C5_SIGMA)
        c5, cnt5 = {}, torch.zeros(1, dtype=torch.int64, device=dev)
        for m5 in C5_LENGTHS:
            p5 = S.corpus_patterns(m5, C5_PATTERNS, PAT_SEED + 2, C5_SIGMA, TEXT_SEED, shard, 2)
            wm5 = S.WmTables.from_patterns(p5, m5, C5_PATTERNS, C5_SIGMA)
            ms5 = sorted(timed(lambda: wm5.scan_device(text5.data_ptr(), shard, cnt5.data_ptr(), S.VARIANT_TUNED, stream), 5, cnt5))[2]
            c5["m%d" % m5] = dict(kernel_ms=round(ms5, 4), **rate(shard, ms5), matches=int(cnt5.item()))
            verify.append(("wm_ascii.m%d" % m5, "wm", p5, m5, C5_PATTERNS, C5_SIGMA, text5, shard, int(cnt5.item())))
            del wm5
        out["wm_ascii"] = dict(workload="WM: %d MiB of 256-symbol text (one GPU's byte range of BASELINE configs[4]), 100000 "
                                        "patterns per set, m=5/12/20" % args.shard_mib, **c5)

    # ---- CPU baselines + bit-exact verification of every count above (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = Cpu()
        sample = min(args.cpu_sample_mib << 20, per_gpu)
        host_text = text[:per_gpu].cpu().numpy()
        prefix = S.corpus_text(4096, TEXT_SEED, SIGMA, offset=0)
        assert np.array_equal(prefix, host_text[:4096]), "device and host corpus generators differ"
        prefix = host_text[:sample]
        # serial search_ac on the prefix
        secs, cpu_counts = cpu.ac_serial(pats, AC_PATTERNS, SIGMA, prefix)
        out["cpu_baseline"] = dict(value=round(8.0 * sample * len(pats) / secs / 1e9, 4), unit="Gbit/s", cores=1, kind=cpu.kind,
                                   cpu=cpu.model,
                                   sample="search_ac (ac/ac.c:198-222) over the first %d MiB of the same text, m=%s, %d patterns "
                                          "each, 1 thread, %.1f s" % (sample >> 20, "/".join(str(m) for m in pats), AC_PATTERNS, secs))
        gpu_counts, c1 = {}, torch.zeros(1, dtype=torch.int64, device=dev)
        for m in AC_LENGTHS:
            c1.zero_()
            acs[m].scan_device(text.data_ptr(), sample, c1.data_ptr(), S.VARIANT_TUNED, stream)
            torch.cuda.synchronize()
            gpu_counts[m] = int(c1.item())
        parity_ok = all(gpu_counts[m] == cpu_counts[m] for m in AC_LENGTHS)
        out["parity"] = dict(bit_exact=parity_ok, gpu_counts=gpu_counts, cpu_counts=cpu_counts, sample_bytes=sample)
        # serial search_wu2 on its (smaller) prefix: the 3-symbol SHIFT table is all zero on DNA from ~1000 patterns up,
        # so every column scans a bucket (0.036 Gbit/s on one thread)
        if wpat is not None:
            wsample = min(args.cpu_wm_sample_mib << 20, per_gpu)
            wsecs, wcnt = cpu.wm_serial(wpat, WM_LENGTH, WM_PATTERNS, SIGMA, host_text[:wsample])
            c1.zero_()
            wm.scan_device(text.data_ptr(), wsample, c1.data_ptr(), S.VARIANT_TUNED, stream)
            torch.cuda.synchronize()
            out["cpu_baseline_wm"] = dict(value=round(8.0 * wsample / wsecs / 1e9, 4), unit="Gbit/s", cores=1, kind=cpu.kind,
                                          cpu=cpu.model, counts_match=int(c1.item()) == wcnt,
                                          sample="search_wu2 (wu/wu.c:151-209) over the first %d MiB of the same text, %d patterns of "
                                                 "length %d, 1 thread, %.1f s" % (wsample >> 20, WM_PATTERNS, WM_LENGTH, wsecs))
            parity_ok = parity_ok and out["cpu_baseline_wm"]["counts_match"]
        # the legacy host-pointer path (search_ac): device allocation + H2D copy + kernel, PCIe-bound; never `value`
        t0 = time.perf_counter()
        legacy_cnt, _ = acs[AC_LENGTHS[0]].count_host(prefix, S.VARIANT_TUNED)
        secs = time.perf_counter() - t0
        out["host_pointer_path"] = dict(what="smh_ac_count_host (what search_ac runs) on the same %d MiB sample: hipMalloc + "
                                             "pageable H2D copy + kernel + D2H of the count" % (sample >> 20),
                                        GBps=round(sample / secs / 1e9, 2), seconds=round(secs, 4),
                                        count_matches=legacy_cnt == cpu_counts[AC_LENGTHS[0]])
        if cpu.kind == "reference":
            secs, ok, wall = cpu.ac_all_cores_reference(pats, AC_PATTERNS, SIGMA, prefix, cpu_counts)
            allc = dict(value=round(8.0 * sample * len(pats) / secs / 1e9, 3), unit="Gbit/s", cores=cpu.cores, kind="reference",
                        cpu=cpu.model, counts_match=ok,
                        sample="same sample as byte-range shards (main.c:467-477) on %d threads; time = slowest shard's search_ac "
                               "per set, summed (%.2f s; %.1f s wall with preproc_ac repeated per shard as every MPI rank of the "
                               "reference does)" % (cpu.cores, secs, wall))
            if wpat is not None:
                wsecs, wok = cpu.wm_all_cores_reference(wpat, WM_LENGTH, WM_PATTERNS, SIGMA, host_text[:wsample], wcnt)
                allc["wm"] = dict(value=round(8.0 * wsample / wsecs / 1e9, 3), unit="Gbit/s", counts_match=wok,
                                  sample="search_wu2, the WM sample as %d byte-range shards, slowest shard %.2f s" % (cpu.cores, wsecs))
                ok = ok and wok
            out["cpu_baseline_all_cores"] = allc
            parity_ok = parity_ok and ok
        # every `matches` on this line against a CPU count of the SAME text (full length), all cores
        t0 = time.perf_counter()
        verified, host_cache = {}, {id(text): host_text}
        for name, algo, pat, m, p, sigma, dtext, n, got in verify:
            if id(dtext) not in host_cache:
                host_cache = {id(text): host_text, id(dtext): dtext[:n].cpu().numpy()}  # one 4 GiB copy at a time
            h = host_cache[id(dtext)][:n]
            want = cpu.ac_count(pat, m, p, sigma, h) if algo == "ac" else cpu.wm_count(pat, m, p, sigma, h)
            verified[name] = dict(gpu=got, cpu=int(want), equal=int(want) == got, text_bytes=n)
            parity_ok = parity_ok and int(want) == got
        if mixed is not None:  # the mixed-length set: sum over its 25 length classes of the restated search_ac, full text
            want = 0
            for L in range(8, 33):
                flat = mixed[0][sum(mixed[1][:(L - 8) * 40]):sum(mixed[1][:(L - 8) * 40]) + 40 * L]
                want += cpu.ac_count(flat, L, 40, SIGMA, host_text)
            for name in ("ac", "wm"):
                got = out["mixed_8_32"][name]["matches"]
                verified["mixed_8_32." + name] = dict(gpu=got, cpu=int(want), equal=int(want) == got, text_bytes=per_gpu)
                parity_ok = parity_ok and int(want) == got
        out["verified"] = dict(checker="restated search_ac / search_wu2 (oracle/, pinned to the reference on the golden vectors) over "
                                       "byte-range shards with an m-1 halo on %d threads; full text of every configuration" % cpu.cores,
                               seconds=round(time.perf_counter() - t0, 1), all_equal=all(v["equal"] for v in verified.values()),
                               counts=verified)
        if not parity_ok:
            print(json.dumps(out))
            raise SystemExit("PARITY FAILURE: GPU counts differ from the CPU reference")

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
