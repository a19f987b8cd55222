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

Extra objects on the JSON line (every one of them at every N unless it says otherwise):
  roofline               HBM bound; achieved = algorithmic bytes per launch (1 byte per text symbol) / mean
                         launch duration from events on the launch stream; traffic from the committed
                         rocprofv3 --pmc passes when they were taken on THIS build of the kernels
  ac / wm                per-configuration rates (WM = BASELINE configs[2]: same text, 10 000 x m=8)
  ac_8000_patterns       BASELINE configs[3]: AC, 8 000 patterns, m = 8/16/32, a 4 GiB byte range (+ halo) of
                         one (N x 4 GiB) DNA text PER GPU, generated on its device; 64-bit counts all-reduced;
                         aggregate Gbit/s = all bytes / the slowest device's kernel time, per-GPU hbm_frac
  wm_ascii               BASELINE configs[4]: WM, 256-symbol text, 100 000 patterns, m = 5/8/12/20, same sharding
  smh_multi              the same three workloads driven by ONE process through the native C path
                         (csrc/smh_multi.hip: one thread per device for the uploads, kernels side by side,
                         ncclAllReduce(uint64) over an ncclCommInitAll communicator); totals must equal the
                         per-rank path's
  verified               EVERY `matches` above against a CPU count of the same text in the same run: the
                         restated search_ac / search_wu2 (oracle/, pinned to the reference) on the host threads
                         of the rank that owns the shard (cores / N each).  N = 1: the full text of every
                         configuration.  N > 1: the full 1 GiB shard of the headline sets; for the 4 GiB shards
                         a stated slice (--verify-mib: head + the last 64 MiB with the halo).  N per-GPU
                         entries per name; any mismatch fails the run on rank 0.
  cpu_baseline*          N = 1 only (rank 0): the reference's own compiled search_ac / search_wu2 (oracle/_ref;
                         the oracle port when absent), one thread and all cores, on bounded prefixes
  stream_read            what a pure streaming read of the same 1 GiB reaches in this run (best of five variants)
  (the LAST stdout line is a compact record of < 4 KB -- metric, value, roofline, cpu_baseline, one number per configuration;
   everything below is in bench_detail.json next to this file, path + sha256 on the line: compact_line / emit)
  skewed                 N = 1 only (round 4; round 5: with the key table and the window-hash filter among the engines): the BASELINE pattern shapes on text that is NOT i.i.d. uniform -- a genome-like
                         DNA text with repeats / tandem repeats / poly-A runs, a protein-like 20-symbol text, a natural-language-
                         like 256-symbol text, and a text in which one planted pattern recurs every 64 columns (csrc/corpus_gen.h;
                         the reference's own data: main.c:39-109), patterns sampled from those texts.  Per set: the entry point as
                         compiled (the adaptive engine, csrc/smh_runtime.hip, after a few launches), every engine forced,
                         chosen_vs_best_forced, all counts verified against the restated search_ac over the full text
  table_kernels          N = 1 only: the table-walking kernels behind cuda_ac1/2, cuda_wm1/2, cuda_sh1/2,
                         cuda_sbom1/2, cuda_sog1/2 on a 64 MiB prefix (latency-bound by design: the reference's
                         tables walked as given)
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

CONDITION_MS = 25.0  # device time under load before anything is timed (bench.py `conditioned`)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
TEXT_SEED, PAT_SEED, SIGMA = 42, 7, 4
AC_LENGTHS = (8, 16, 32)
AC_PATTERNS = 1000
WM_PATTERNS, WM_LENGTH = 10000, 8
C4_PATTERNS = 8000
C5_PATTERNS, C5_LENGTHS, C5_SIGMA = 100000, (5, 8, 12, 20), 256
C5_MORE_LENGTHS = (6, 7, 9, 10, 16)  # the form boundaries of the byte-gram filters (flat <= 9 / hashed above): kernel time + slice verification


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


def ac_kernel_name(info):
    """Prefix of the kernel instance (as rocprofv3 prints it) that serves the Aho-Corasick entry point for the handle `info`
    (smh_ac_info) describes."""
    if info.scan_dense:  # the dense plan: the pair lookup kernel with the automaton's accepting-bit set
        return "wm_pair_kernel<false, 1024>"
    if info.scan_engine == 1:  # SMH_ALGO_WM: the pair-gram filter scans (ac_host.c, end of the compile); STG = halo staged / 16
        # verify stage: 1 = in registers (STG 5 / 6), 2 = windows from L2 (STG 3 / 4), 0 = staged (STG 1 / 2): smh_wm_info.verify_in_registers
        return "wm_gram_kernel<%d, false, %d, false>" % (info.gram_kind, (1 if info.m <= 17 else 2) + {0: 0, 1: 4, 2: 2}[int(info.verify_in_registers)])
    halo = info.scan_depth - 1
    hc = 1 if halo <= 16 else (2 if halo <= 32 else 4)
    entry = "unsigned short" if (info.scan_stride == 2 or info.lds_rows <= 32768) else "unsigned int"
    # template value of the stride: the hybrid image runs as 4 (full-row lookup left out of range, device probed) or 3 (clamped)
    stride = 4 if info.scan_full_rows else info.scan_stride
    return "ac_dfa_kernel<%s, 4, %d, %d, %s," % (entry, stride, hc, "true" if info.scan_exact else "false")


def measured_traffic(info, nbytes):
    """-> (HBM bytes per launch or None, source string) for the kernel instance that scans the set `info` (smh_ac_info)
    describes over `nbytes` of text.  bench.py cannot read PMC counters itself; profiles/hbm_traffic.json is produced from
    this same command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/collect_counters.sh; FETCH_SIZE doubled per
    the gfx950 correction of MI355X_MICROARCH.md) and carries the build id it was taken on."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "no committed counter pass"
    rec = json.load(open(path))
    have, want = rec.get("build_id"), kernel_build_id()
    if have != want:
        return None, "profiles/hbm_traffic.json was taken on build %s, this is build %s: not quoted" % (have, want)
    prefix = ac_kernel_name(info)
    prefixes = (prefix, prefix.replace(", 4, 4,", ", 4, 3,")) if info.scan_full_rows else (prefix,)
    for pre in prefixes:
        for name, k in rec.get("kernels", {}).items():
            if name.startswith(pre) and ", true, 1024>" not in name:  # "..., true, 1024>" = the positions-mode instance
                # an instance that also served the 4 GiB shards: the group of dispatches whose volume is this text's
                groups = [g["hbm_read_bytes"] + k.get("hbm_write_bytes", 0) for g in k.get("by_text_size", [])] or [k["hbm_bytes"]]
                best = min(groups, key=lambda v: abs(v - nbytes))
                if not 0.9 * nbytes < best < 2.0 * nbytes:
                    continue
                return best, "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, build %s, %s)" % (
                    have, rec.get("profile", "?"))
    return None, "no counter pass for " + prefix


LINE_LIMIT = 4096  # the driver keeps an 8 KB tail of stdout and parses the LAST line: the full record goes to a side file


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(out, detail_path=None, detail_sha=None):
    """The ONE short JSON line the driver parses (reference: one line per measurement, cuda/cuda_ac.cu:675, main.c:664-670):
    the contract keys, `roofline`, `cpu_baseline`, and one number per BASELINE configuration.  Everything else `out` holds
    stays in bench_detail.json (path + sha256 on the line).  Always shorter than LINE_LIMIT: optional groups are dropped
    from the end until it is (tests/test_bench_helpers.py)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    cfg = dict(out.get("config", {}))
    cfg["workload"] = str(cfg.get("workload", ""))[:260]
    line["config"] = cfg
    # did the run hold?  (mandatory since round 6: a failed side leg or a parity failure must not hide in an optional group)
    for key in ("parity_ok", "smh_multi_ok", "error"):
        if key in out:
            line[key] = out[key] if key != "error" else str(out[key])[:300]
    w = out.get("world")
    if w:  # who ran: backend and world size as torch.distributed reports them, one card identity per rank
        line["world"] = dict(world_size=w.get("world_size"), backend=w.get("backend"), distinct_cards=w.get("distinct_cards"),
                             rehearsal=w.get("rehearsal"),
                             ranks=[_pick(r, ("rank", "local_rank", "device", "pci_bus_id")) for r in w.get("ranks", [])][:16])
        buses = {r.get("pci_bus_id") for r in w.get("ranks", [])}
        if w.get("distinct_cards") and len(buses) != w.get("distinct_cards"):  # cards told apart by UUID (rank_identity): say so
            line["world"]["ranks"] = [_pick(r, ("rank", "local_rank", "device", "pci_bus_id", "uuid")) for r in w.get("ranks", [])][:16]
    roof = _pick(out.get("roofline", {}), ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_instance", "launch_ms",
                                           "algorithmic_bytes_per_launch", "of_stream_read", "chosen_engine"))
    roof["traffic_source"] = str(out.get("roofline", {}).get("traffic_source", ""))[:140]
    auto = out.get("roofline", {}).get("automaton")
    if auto:
        roof["automaton"] = _pick(auto, ("kernel_instance", "launch_ms", "frac"))
    line["roofline"] = roof
    if "cpu_baseline" in out:
        cb = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu", "host_cpus"))
        cb["sample"] = str(out["cpu_baseline"].get("sample", ""))[:200]
        line["cpu_baseline"] = cb
    optional = []  # (key, value) in order of importance; dropped from the end when the line would be too long
    ver = out.get("verified")
    if ver:
        bad = sorted(k for k, v in ver.get("counts", {}).items() if not v.get("equal"))
        optional.append(("verified", dict(all_equal=ver.get("all_equal"), n=len(ver.get("counts", {})), seconds=ver.get("seconds"), unequal=bad[:8])))
    fr = lambda obj: {k: v["hbm_frac"] for k, v in (obj or {}).items() if isinstance(v, dict) and "hbm_frac" in v}
    frac = {}
    for key in ("ac", "ac_automaton", "wm_long", "ac_8000_patterns", "wm_ascii", "wm_ascii_more"):
        if fr(out.get(key)):
            frac[key] = fr(out[key])
    if isinstance(out.get("wm"), dict) and "hbm_frac" in out["wm"]:
        frac["wm"] = out["wm"]["hbm_frac"]
    if out.get("mixed_8_32"):
        frac["mixed_8_32"] = fr(out["mixed_8_32"])
    if frac:
        optional.append(("hbm_frac", frac))
    if "stream_read" in out:
        optional.append(("stream_read", _pick(out["stream_read"], ("GBps", "hbm_frac"))))
    for key in ("cpu_baseline_wm", "cpu_baseline_all_cores"):
        if key in out:
            optional.append((key, _pick(out[key], ("value", "unit", "cores", "cpu_quota", "kind", "counts_match"))))
    if "host_pointer_path" in out:
        optional.append(("host_pointer_path", _pick(out["host_pointer_path"], ("GBps", "first_call_GBps", "count_matches"))))
    st = out.get("small_text")
    if st:  # the reference's data-set sizes: this size's rate over the 1 GiB rate, [AC 1000 x 8, WM 8000 x 8]
        optional.append(("small_text_of_gib_rate", {k: [v2.get("of_gib_rate") for k2, v2 in v.items() if isinstance(v2, dict)]
                                                    for k, v in st.items() if isinstance(v, dict)}))
    pp = out.get("preproc")
    if pp:
        optional.append(("preproc_s", {k: v.get("preproc_s") for k, v in pp.get("sets", {}).items()}))
    sk = out.get("skewed")
    if sk:
        s = {"worst_chosen_vs_best_forced": sk.get("worst_chosen_vs_best_forced")}
        for cname, cobj in sk.items():
            if isinstance(cobj, dict):
                s[cname] = {k: v["chosen"]["hbm_frac"] for k, v in cobj.items() if isinstance(v, dict) and "chosen" in v}
        optional.append(("skewed", s))
    mg = out.get("smh_multi")
    if mg:
        eq = mg.get("totals_equal_per_rank_path", {})
        optional.append(("smh_multi", dict(devices=mg.get("devices"), totals_equal=bool(eq) and all(eq.values()), error=mg.get("error"))))
    if "after_idle" in out:
        optional.append(("after_idle", _pick(out["after_idle"], ("ms_per_step", "value"))))
    optional.append(("device", out.get("device")))
    optional.append(("kernel_build_id", out.get("kernel_build_id")))
    if out.get("rehearsal"):
        optional.append(("rehearsal", str(out["rehearsal"])[:120]))
    if "wall_s" in out:
        optional.append(("wall_s", out["wall_s"]))
    if "phases_s" in out:
        optional.append(("phases_s", out["phases_s"]))
    if detail_path:
        line["detail"] = dict(path=detail_path, sha256=detail_sha)
    kept = list(optional)
    while True:
        full = dict(line)
        full.update(kept)
        text = json.dumps(full, separators=(",", ":"))
        if len(text) < LINE_LIMIT or not kept:
            return text
        kept.pop()


def emit(out):
    """Full record -> bench_detail.json next to this file (and under gpurun_out/ when that exists, so that it travels back
    from a GPU box); the compact line -> stdout, last."""
    blob = json.dumps(out, indent=1, sort_keys=False)
    sha = hashlib.sha256(blob.encode()).hexdigest()
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for path in paths:
        try:
            with open(path, "w") as f:
                f.write(blob)
            written = written or os.path.relpath(path, ROOT)
        except OSError:
            pass
    sys.stdout.flush()
    print(compact_line(out, written, sha if written else None), flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs' worth of run time the container's cgroup grants this process (cpu.max / cfs quota), None = no limit.  A box of this
    pool shows 256 CPUs and grants 16: 256 threads of search_ac then deliver 16 threads' worth (13.7 Gbit/s = 16 x 0.87)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


# ---------------------------------------------------------------------------------------------------------
# CPU side: the checker.  The only place bench.py touches oracle/ (through tests/oracle_lib.py).
class Cpu:
    def __init__(self, world=1):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from concurrent.futures import ThreadPoolExecutor
        self.O = O
        self.host_threads = len(os.sched_getaffinity(0))   # what this process may run on
        self.host_cpus = os.cpu_count()                    # what the machine has
        # every rank of an N-rank job checks its own shard: an equal share of the threads each, at most 64
        self.cores = max(1, min(self.host_threads // max(world, 1), 64))
        self.pool = ThreadPoolExecutor(self.cores)
        # the all-cores baseline (N = 1 only) is not capped: every thread this process may run on
        self.all_cores = max(1, self.host_threads)
        self._all_pool = None
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
        from concurrent.futures import ThreadPoolExecutor
        self._all_pool = self._all_pool or ThreadPoolExecutor(self.all_cores)
        for m, pat in pats.items():
            ranges = [O.shard_range(n, self.all_cores, r, m) for r in range(self.all_cores)]
            parts = list(self._all_pool.map(lambda be: O.ref_ac(pat, m, p, sigma, text[be[0]:be[1]]), ranges))
            ok = ok and sum(q[0] for q in parts) == want[m]
            secs += max(q[3] for q in parts)
        return secs, ok, time.perf_counter() - wall0

    def wm_all_cores_reference(self, pat, m, p, sigma, text, want):
        O, n = self.O, len(text)
        from concurrent.futures import ThreadPoolExecutor
        self._all_pool = self._all_pool or ThreadPoolExecutor(self.all_cores)
        ranges = [O.shard_range(n, self.all_cores, r, m) for r in range(self.all_cores)]
        parts = list(self._all_pool.map(lambda be: O.ref_wu(pat, m, p, sigma, text[be[0]:be[1]], flat=True), ranges))
        return max(q[3] for q in parts), sum(q[0] for q in parts) == want

    # --- verification (restated search, tables built once and shared by the threads) ---
    def counter(self, algo, pat, m, p, sigma):
        """-> count(host text) over byte-range pieces with an m-1 halo on this rank's threads"""
        O = self.O
        if algo == "ac":
            _, tabs = O.oracle_ac(pat, m, p, sigma)
            one = lambda t: O.oracle_ac_search_tables(t, sigma, tabs)
        else:
            csr = O.WMTablesCSR(pat, m, p, sigma)
            one = csr.search

        def count(text):
            n = len(text)
            pieces = max(self.cores * 4, 1)
            ranges = [O.shard_range(n, pieces, r, m) for r in range(pieces)]
            return int(sum(self.pool.map(lambda be: one(text[be[0]:be[1]]), ranges)))
        return count


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


# ---------------------------------------------------------------------------------------------------------
# The native one-process leg: `bench.py --multi-leg N ...` is started by rank 0 as a CHILD process (torch-free:
# ctypes over libsmatcher_hip.so only) while the ranks keep their GPUs idle, and prints one JSON object.
def multi_leg(args):
    import smatcher_hip as S
    n_dev = args.multi_leg
    share = os.environ.get("SMH_MULTI_SHARE_DEVICE", "0") not in ("", "0")  # one-card rehearsal of the N-device flow (csrc/smh_multi.hip)
    if S.device_count() < n_dev and not share:
        print(json.dumps({"error": "%d device(s) visible to the one-process leg, %d wanted" % (S.device_count(), n_dev)}))
        return
    mg = S.MultiGpu(n_dev)
    per_gpu, shard = args.mib_per_gpu << 20, args.shard_mib << 20
    out = {"devices": n_dev, "reduce": "ncclAllReduce(uint64, sum) over ncclCommInitAll" if mg.uses_rccl else "host sum",
           "rehearsal": "SMH_MULTI_SHARE_DEVICE: %d logical shards on %d card(s); rates are not N-GPU rates" % (n_dev, S.device_count()) if share else None,
           "what": "ONE process drives all devices through smh_multi_* (csrc/smh_multi.hip); seconds = launches on every "
                   "device + the count all-reduce + read-back, table sets prepared before the clock; timed after %.0f ms of "
                   "back-to-back count calls (the per-rank path's conditioning)" % CONDITION_MS}

    def run(name, algo, sigma, lengths, p, seed, n_each, workload):
        n_total = n_each * n_dev
        mg.generate_text(n_total, TEXT_SEED, sigma, max(lengths) - 1)
        obj = {"workload": workload}
        for m in lengths:
            pat = S.corpus_patterns(m, p, seed, sigma, TEXT_SEED, n_total, 2)
            h = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
            count = mg.ac_count if algo == "ac" else mg.wm_count
            t0 = time.perf_counter()
            mg.prepare(h)
            prep = time.perf_counter() - t0
            # the same steady state the per-rank path measures at (`conditioned` in main): count calls back to back for
            # CONDITION_MS before the timed ones -- the first milliseconds after an idle gap run 15-20 % slower
            first = count(h)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < CONDITION_MS * 1e-3:
                count(h)
            runs = [first] + [count(h) for _ in range(args.steps)]
            first_call = runs.pop(0)[2]
            secs = sorted(r[2] for r in runs)
            total, per = runs[-1][0], runs[-1][1]
            assert all(r[0] == total for r in runs)
            med = secs[len(secs) // 2]
            gbs = n_total / med / 1e9
            obj["m%d" % m] = dict(seconds=round(med, 6), first_call_seconds=round(first_call, 6), min_seconds=round(secs[0], 6),
                                  prepare_seconds=round(prep, 4), GBps=round(gbs, 1), Gbit_s=round(8 * gbs, 1),
                                  hbm_frac_per_gpu=round(gbs / n_dev / HBM_PEAK_GBS, 4), matches=total, per_gpu_matches=per)
            h.close()
        out[name] = obj

    run("ac", "ac", SIGMA, AC_LENGTHS, AC_PATTERNS, PAT_SEED, per_gpu,
        "BASELINE configs[1] shape: %d MiB of DNA per device, %d patterns, m=8/16/32" % (args.mib_per_gpu, AC_PATTERNS))
    if not args.no_wm:
        run("ac_8000_patterns", "ac", SIGMA, AC_LENGTHS, C4_PATTERNS, PAT_SEED + 3, shard,
            "BASELINE configs[3]: %d MiB of DNA per device, 8000 patterns" % args.shard_mib)
        run("wm_ascii", "wm", C5_SIGMA, C5_LENGTHS, C5_PATTERNS, PAT_SEED + 2, shard,
            "BASELINE configs[4]: %d MiB of 256-symbol text per device, 100000 patterns" % args.shard_mib)
    mg.close()
    print(json.dumps(out))


# ---------------------------------------------------------------------------------------------------------
# The reference-shaped host calls (include/smatcher.h), as main.c makes them: caller-owned tables, initialised by the caller.
def time_preproc_ac(S, np, pat, m, p, sigma):
    """seconds of preproc_ac (ac/ac.c:224-245) into caller-owned arrays initialised as main.c:410-420 does"""
    rows = m * p + 1
    st = np.full(rows * sigma, -1, dtype=np.int32)
    supply, final = np.zeros(rows, dtype=np.uint32), np.zeros(rows, dtype=np.uint32)
    padded = np.zeros((p, m + 1), dtype=np.uint8)  # every pattern buffer m + 1 bytes (ac_addstring reads one past)
    padded[:, :m] = np.asarray(pat, dtype=np.uint8).reshape(p, m)
    base, stride = padded.ctypes.data, m + 1
    arr = (S.u8p * p)(*[C.cast(base + j * stride, S.u8p) for j in range(p)])
    t0 = time.perf_counter()
    tab = S.lib.preproc_ac(arr, m, p, sigma, st.ctypes.data_as(S.i32p), supply.ctypes.data_as(S.u32p), final.ctypes.data_as(S.u32p))
    secs = time.perf_counter() - t0
    S.lib.free_ac(tab, sigma)
    return secs


def time_preproc_wu2(S, np, pat, m, p, sigma):
    """-> (seconds of preproc_wu2 (wu/wu.c:211-251), the tables) -- tables allocated and initialised as main.c:429-449 does"""
    S.lib.wu_determine_shiftsize(sigma)
    ss = S.shiftsize_global()
    shift = np.full(ss, m - 3 + 1, dtype=np.int32)
    pv, pi, ps = np.zeros(ss * p, dtype=np.int32), np.zeros(ss * p, dtype=np.int32), np.zeros(ss, dtype=np.int32)
    flat = np.ascontiguousarray(pat, dtype=np.uint8)
    ptrs = [a.ctypes.data_as(S.i32p) for a in (shift, pv, pi, ps)]
    t0 = time.perf_counter()
    S.lib.preproc_wu2(flat.ctypes.data_as(S.u8p), m, p, sigma, 3, *ptrs)
    return time.perf_counter() - t0, (flat, shift, pv, pi, ps)


def time_cuda_wm_calls(S, np, pat, m, p, sigma, host_text):
    """cuda_wm1..5 back to back on one set of caller tables, as main.c:623-648: wall seconds of every call"""
    _, (flat, shift, pv, pi, ps) = time_preproc_wu2(S, np, pat, m, p, sigma)
    ptrs = [a.ctypes.data_as(S.i32p) for a in (shift, pv, pi, ps)]
    text = np.ascontiguousarray(host_text, dtype=np.uint8)
    secs, counts = [], []
    for k in range(1, 6):
        fn = getattr(S.lib, "cuda_wm%d" % k)
        gpu_time = C.c_double(0)
        t0 = time.perf_counter()
        counts.append(int(fn(flat.ctypes.data_as(S.u8p), m, text.ctypes.data_as(S.u8p), len(text), p, sigma, 3, *ptrs, C.byref(gpu_time))))
        secs.append(time.perf_counter() - t0)
    if len(set(counts)) != 1:
        raise SystemExit("PARITY FAILURE: cuda_wm1..5 disagree: %r" % counts)
    return secs



# ---------------------------------------------------------------------------------------------------------
# The per-rank run.  `Run` holds what the phases share (arguments, device, the headline's text and handles, the record
# `out` on rank 0, the list `verify` of every count the CPU recounts at the end); each phase below is one function.
class Run:
    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        import smatcher_hip as S
        import sharded
        self.args, self.np, self.torch, self.dist, self.S, self.sharded = args, np, torch, dist, S, sharded
        self.wall_t0 = time.perf_counter()
        self.phases, self._phase_t = {}, self.wall_t0
        world_env = os.environ.get("WORLD_SIZE")
        self.world = int(world_env or "1")
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("WORLD_SIZE %d != --gpus %d" % (self.world, args.gpus))
        if not torch.cuda.is_available() or S.device_count() < 1:
            raise SystemExit("bench.py needs a HIP device: the scan path has no CPU fallback")
        if self.local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d device(s) visible" % (self.rank, self.local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.backend = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.share_device:
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            self.backend = dist.get_backend()
        self.per_gpu = args.mib_per_gpu << 20
        self.n_total = self.per_gpu * self.world
        self.shard = args.shard_mib << 20
        # N > 1: what a rank's share of the host threads recounts in about a minute per 32 GB configuration -- 8 MiB per thread,
        # 128..512 MiB of every 4 GiB shard (6 ranks on one box: 153 s of wall at 512 MiB, 149 s of it this; profiles/r05_final/rehearse6*)
        # (threads the container's CPU quota does not back recount nothing: a box of this pool shows 256 CPUs and grants 16)
        eff_cpus = len(os.sched_getaffinity(0)) if cpu_quota() is None else min(len(os.sched_getaffinity(0)), max(1, int(cpu_quota() + 0.5)))
        auto_mib = max(128, min(512, 8 * max(1, eff_cpus // max(self.world, 1))))
        self.verify_budget = (args.verify_mib << 20) if args.verify_mib >= 0 else (0 if self.world == 1 else auto_mib << 20)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.out = None      # rank 0: the record
        self.verify = []     # every rank: (name, algorithm, patterns, m, p, sigma, device text, shard length, gpu count, scan(ptr, n) -> count or None)
        self.one = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.parity_ok = True
        self.multi_ok = None  # None: the leg was not asked for
        self.wm = self.wpat = self.mixed = None

    # ---- helpers ----
    def mark(self, name):  # wall seconds of the phase that just ended (rank 0's clock)
        now = time.perf_counter()
        self.phases[name] = round(self.phases.get(name, 0.0) + now - self._phase_t, 1)
        self._phase_t = now

    def ev(self):
        return self.torch.cuda.Event(enable_timing=True)

    def corpus(self, n, offset, sigma):
        t = self.torch.empty(n + 64, dtype=self.torch.uint8, device=self.dev)
        rc = self.S.lib.smh_corpus_text_device(C.c_void_p(t.data_ptr()), n, offset, TEXT_SEED, sigma, C.c_void_p(self.stream))
        if rc != 0:
            raise SystemExit("corpus generation failed: " + self.S.lib.smh_last_error().decode())
        return t

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def conditioned(self, launch, est_ms, ms=CONDITION_MS):
        """Launches `launch` back to back for about `ms` of device time, no synchronisation behind it.  After an idle gap
        (a torch.cuda.synchronize(), the CPU legs) the device runs ~13 launches at its steady rate, then 3-9 ms into the
        load every kernel takes 15-20 % longer for a few milliseconds, then settles again (profiles/r03_s/exp_sustain.log:
        per-launch times of 60-300 back-to-back scans): a transient of the power management, not of the kernels.  The
        driver's --warmup 5 covers the first 2.6 ms of it and put the hump inside the timed steps.  Every measurement of this
        file therefore starts on a device that has been under the same load for CONDITION_MS."""
        for _ in range(max(2, int(ms / max(est_ms, 1e-3)) + 1)):
            launch()

    def timed(self, launch, reps, counter):
        """`reps` launches bracketed by events on the launch stream -> list of ms"""
        a0, b0 = self.ev(), self.ev()
        a0.record()
        launch()
        b0.record()
        self.torch.cuda.synchronize()
        self.conditioned(launch, a0.elapsed_time(b0))
        evs = [(self.ev(), self.ev()) for _ in range(reps)]
        for a, b in evs:
            counter.zero_()
            a.record()
            launch()
            b.record()
        self.torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    @staticmethod
    def rate(nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return dict(GBps=round(gbs, 1), Gbit_s=round(8 * gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4))

    def scan_with(self, handle):
        def scan(ptr, n):
            self.one.zero_()
            handle.scan_device(ptr, n, self.one.data_ptr(), self.S.VARIANT_TUNED, self.stream)
            self.torch.cuda.synchronize()
            return int(self.one.item())
        return scan

    def sharded_set(self, name, algo, pat, m, p, sigma, dtext, n_m, reps, engine=None):
        """one pattern set over every rank's shard of a sharded text: per-rank kernel time (events), counts all-reduced"""
        S, torch, sharded = self.S, self.torch, self.sharded
        handle = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
        if engine is not None:
            handle.set_scan_engine(engine)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.dev)
        handle.scan_device(dtext.data_ptr(), n_m, cnt.data_ptr(), S.VARIANT_TUNED, self.stream)  # tables up, code loaded
        self.barrier()  # the ranks' launches run side by side, as in the job
        ms = sorted(self.timed(lambda: handle.scan_device(dtext.data_ptr(), n_m, cnt.data_ptr(), S.VARIANT_TUNED, self.stream), reps, cnt))
        local = int(cnt.item())
        sharded.reduce_count(cnt)  # the MPI_Reduce of main.c:656
        recs = sharded.gather_objects(dict(n=n_m, ms=ms[len(ms) // 2], matches=local))
        self.verify.append((name, algo, pat, m, p, sigma, dtext, n_m, local, self.scan_with(handle)))
        obj = sharded.summarize_shard_runs(recs, HBM_PEAK_GBS)
        assert obj["matches"] == int(cnt.item()), "all-reduced count differs from the sum of the gathered shard counts"
        return obj, handle


def mean(xs):
    return sum(xs) / len(xs) if xs else 0.0


def rank_identity(R):
    """Who ran: one {rank, local_rank, device, pci_bus_id, host} per rank, gathered -- what lets a reader of an N > 1 record
    check that N distinct cards took part (main.c:327-333 prints the rank count and nothing else).  Ranks that share a card are an
    error unless the run is the --share-device rehearsal."""
    S, torch = R.S, R.torch
    props = torch.cuda.get_device_properties(R.local_rank)
    mine = dict(rank=R.rank, local_rank=R.local_rank, device=int(torch.cuda.current_device()), pci_bus_id=S.device_pci_bus_id(),
                uuid=str(getattr(props, "uuid", "")), host=socket.gethostname(), pid=os.getpid())
    ranks = R.sharded.gather_objects(mine)
    # a card = (host, PCI bus id, device UUID): two ranks share one only when all three agree (partitioned devices may report
    # one bus id for several logical devices; their UUIDs differ)
    cards = {(r["host"], r["pci_bus_id"], r["uuid"]) for r in ranks}
    if len(cards) != len(ranks) and not R.args.share_device:
        raise SystemExit("bench.py: %d ranks on %d distinct cards %r -- every rank needs a card of its own (--share-device rehearses the "
                         "flow on one card)" % (len(ranks), len(cards), sorted(cards)))
    return ranks, len(cards)


def sharding_label(world, backend, share_device):
    """config.sharding: the decomposition and the collective that actually summed the counts"""
    if world == 1:
        return "one rank: the whole text, no collective"
    how = {"nccl": "RCCL (torch.distributed backend nccl) all-reduce of the 64-bit counts",
           "gloo": "gloo all-reduce of the 64-bit counts (host)"}.get(backend, "%s all-reduce of the 64-bit counts" % backend)
    return "byte-range x%d, m-1 halo; %s%s" % (world, how, "; REHEARSAL: every rank on one card" if share_device else "")


# ---------------------------------------------------------------------------------------------------------
def phase_headline(R):
    """BASELINE configs[1]: the three AC sets over this rank's 1 GiB, W warm-up + K timed steps -> `value`, `roofline`"""
    args, S, torch, sharded = R.args, R.S, R.torch, R.sharded
    R.mark('start (imports, process group)')
    # ---- pattern sets (host) and compiled automata; what the reference times as preproc (main.c:246-262) is timed too
    R.pats = {m: S.corpus_patterns(m, AC_PATTERNS, PAT_SEED, SIGMA, TEXT_SEED, R.n_total, 2) for m in AC_LENGTHS}
    R.compile_s = {}
    R.acs = {}
    for m in AC_LENGTHS:
        t0 = time.perf_counter()
        R.acs[m] = S.AcAutomaton.from_patterns(R.pats[m], m, AC_PATTERNS, SIGMA)
        R.compile_s[m] = time.perf_counter() - t0

    # ---- this rank's byte range of the N GiB text, generated in HBM (never crosses PCIe)
    begin, n_alloc, R.shard_len = sharded.shard_plan(R.n_total, R.world, R.rank, AC_LENGTHS)
    assert begin == R.rank * R.per_gpu
    R.text = R.corpus(n_alloc, begin, SIGMA)
    torch.cuda.synchronize()
    text, acs, shard_len, stream = R.text, R.acs, R.shard_len, R.stream

    # every step has its own count buffer: its all-reduce is started behind its three scans and runs on RCCL's stream
    # while the next step's scans run on ours; all of them are waited for inside the timed region
    step_counts = torch.zeros((2 * (args.warmup + args.steps) + 1, len(AC_LENGTHS)), dtype=torch.int64, device=R.dev)

    def step(k, events=None):
        c = step_counts[k]  # zero since its allocation: every step accumulates into a row of its own
        if events is not None:
            events[0].record()  # one event between consecutive launches: the end of one is the start of the next
        for i, m in enumerate(AC_LENGTHS):
            acs[m].scan_device(text.data_ptr(), shard_len[m], c.data_ptr() + 8 * i, S.VARIANT_TUNED, stream)
            if events is not None:
                events[i + 1].record()
        return sharded.reduce_count_async(c)

    # steady state first (see `conditioned`): the step's scans without the reduce, into the scratch row, for CONDITION_MS
    def scans_only():
        for i, m in enumerate(AC_LENGTHS):
            acs[m].scan_device(text.data_ptr(), shard_len[m], step_counts[-1].data_ptr() + 8 * i, S.VARIANT_TUNED, stream)
    t0 = time.perf_counter()
    scans_only()  # the handles' first launches: table sets go up (blocking), code objects load
    torch.cuda.synchronize()
    R.first_scans_s = time.perf_counter() - t0
    R.barrier()
    R.conditioned(scans_only, 0.6)
    sharded.finish([step(k) for k in range(args.warmup)])
    evs = [[R.ev() for _ in range(len(AC_LENGTHS) + 1)] for _ in range(args.steps)]
    R.barrier()
    t0 = time.perf_counter()
    sharded.finish([step(args.warmup + k, evs[k]) for k in range(args.steps)])
    R.barrier()
    elapsed = time.perf_counter() - t0
    counts = step_counts[args.warmup + args.steps - 1] if args.steps else step_counts[0]
    elapsed = max(sharded.gather_objects(elapsed))  # the job's time is the slowest rank's
    # for the record: the same W + K steps started on an IDLE device, i.e. what this file measured before it conditioned
    # the device (the transient of `conditioned` falls into the timed steps)
    time.sleep(0.3)
    base = args.warmup + args.steps
    sharded.finish([step(base + k) for k in range(args.warmup)])
    R.barrier()
    t0 = time.perf_counter()
    sharded.finish([step(base + args.warmup + k) for k in range(args.steps)])
    R.barrier()
    idle_elapsed = max(sharded.gather_objects(time.perf_counter() - t0))
    total_counts = [int(x) for x in counts.tolist()]
    # per-GPU counts for the report: one untimed pass without the reduce, then one small all-gather
    counts = step_counts[-1]
    counts.zero_()
    for i, m in enumerate(AC_LENGTHS):
        acs[m].scan_device(text.data_ptr(), shard_len[m], counts.data_ptr() + 8 * i, S.VARIANT_TUNED, stream)
    torch.cuda.synchronize()
    per_gpu_counts = sharded.gather_counts(counts).tolist()
    R.local_counts = [int(x) for x in counts.tolist()]

    # per-launch durations (ms) from the events on the launch stream
    R.kern_ms = kern_ms = {m: [evs[k][i].elapsed_time(evs[k][i + 1]) for k in range(args.steps)] for i, m in enumerate(AC_LENGTHS)}
    bits_per_step = 8.0 * sum(sum(sl.values()) for sl in sharded.gather_objects(shard_len))
    value = bits_per_step * args.steps / elapsed / 1e9
    all_kern_ms = sharded.gather_objects({m: mean(kern_ms[m]) for m in AC_LENGTHS})  # [rank][m]
    ranks, n_cards = rank_identity(R)
    R.mark('headline (compile, text, timed steps)')

    for i, m in enumerate(AC_LENGTHS):
        R.verify.append(("ac.m%d" % m, "ac", R.pats[m], m, AC_PATTERNS, SIGMA, text, shard_len[m], R.local_counts[i], R.scan_with(acs[m])))
    R.dom = max(AC_LENGTHS, key=lambda m: mean(kern_ms[m]))
    if R.rank != 0:
        return
    ac_detail = {}
    for i, m in enumerate(AC_LENGTHS):
        info = acs[m].info()
        ms = mean(kern_ms[m])
        ac_detail["m%d" % m] = dict(kernel_ms=round(ms, 4), median_ms=round(sorted(kern_ms[m])[len(kern_ms[m]) // 2], 4),
                                    min_ms=round(min(kern_ms[m]), 4), **R.rate(shard_len[m], ms),
                                    per_gpu_ms=[round(r[m], 4) for r in all_kern_ms],
                                    dfa_rows=info.rows, lds_rows=info.lds_rows, lds_bytes=info.lds_bytes,
                                    scan_stride=info.scan_stride, scan_depth=info.scan_depth,
                                    scan_exact=info.scan_exact, scan_full_rows=info.scan_full_rows, scan_dense=info.scan_dense,
                                    kernel_instance=ac_kernel_name(info),
                                    scan_engine="suffix-filter kernels" if info.scan_engine == S.ALGO_WM else "automaton kernels",
                                    matches=total_counts[i])
    dom = R.dom
    dom_ms = mean(kern_ms[dom])
    achieved = shard_len[dom] / (dom_ms * 1e-3) / 1e9
    traffic, traffic_source = measured_traffic(acs[dom].info(), shard_len[dom])
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                    kernel="%s (m=%d set)" % (ac_kernel_name(acs[dom].info()).split("<")[0], dom),
                    kernel_instance=ac_kernel_name(acs[dom].info()), launch_ms=round(dom_ms, 4),
                    algorithmic_bytes_per_launch=shard_len[dom])
    R.out = {
        "metric": "Gbit/s text scanned (AC and WM) at 1/2/4/8 MI355X; % HBM roofline",
        "value": round(value, 2), "unit": "Gbit/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "conditioning": "the step's scans run back to back for %.0f ms of device time before the %d warm-up steps (and every "
                        "side measurement's launch likewise before its timed repetitions): steady-state clocks -- 3-9 ms after an "
                        "idle gap every kernel runs 15-20 %% slower for a few ms (profiles/r03_s/exp_sustain.log)" % (CONDITION_MS, args.warmup),
        "config": {"workload": "AC on MI355X: %d MiB synthetic DNA text per GPU resident in HBM, %d patterns per set, "
                               "pattern lengths 8-32 as fixed-length sets m=8/16/32 (BASELINE configs[1]); "
                               "step = 3 scans + count all-reduce" % (args.mib_per_gpu, AC_PATTERNS),
                   "text_bytes_per_gpu": R.per_gpu, "alphabet": SIGMA, "patterns": AC_PATTERNS,
                   "pattern_lengths": list(AC_LENGTHS), "text_seed": TEXT_SEED, "pattern_seed": PAT_SEED,
                   "sharding": sharding_label(R.world, R.backend, args.share_device)},
        # who ran (round 6): the backend and world size as torch.distributed reports them, one identity per rank, distinct cards
        "world": dict(world_size=R.dist.get_world_size() if R.world > 1 else 1, backend=R.backend or "none", distinct_cards=n_cards,
                      rehearsal=bool(args.share_device), ranks=ranks),
        "after_idle": {"what": "the same %d warm-up + %d timed steps started on an idle device, without the conditioning: the "
                               "power-management transient falls into the timed steps" % (args.warmup, args.steps),
                       "ms_per_step": round(idle_elapsed / max(args.steps, 1) * 1e3, 4),
                       "value": round(bits_per_step * args.steps / idle_elapsed / 1e9, 2) if args.steps else None},
        "roofline": roofline, "ac": ac_detail, "device": S.device_name(), "kernel_build_id": kernel_build_id(),
        "per_gpu_matches": {"m%d" % m: [int(r[i]) for r in per_gpu_counts] for i, m in enumerate(AC_LENGTHS)},
    }
    if args.share_device:
        R.out["rehearsal"] = "--share-device: every rank on device 0, process group over gloo; rates are not N-GPU rates"


def phase_stream_read_and_positions(R):
    """rank 0: what a pure streaming read of the same 1 GiB reaches on this device, same run -- the best of five
    read-only kernels (tools/readsweep.hip picked their shapes: about 32 KiB in flight per CU streams best) -- and the m = 16
    set's END columns into a device buffer"""
    if R.rank != 0:
        return
    S, torch, out = R.S, R.torch, R.out
    probe = torch.zeros(1, dtype=torch.int64, device=R.dev)
    names = ["grid-stride, 16-byte loads, 32 waves/CU", "4 KiB wave-chunks, 8 waves/CU", "grid-stride, 8 waves/CU",
             "4 KiB wave-chunks, two in flight, 4 waves/CU", "4 KiB wave-chunks from the LDS counter, 16 waves/CU (the scan kernels' shape)"]
    variants = {}
    for v, nm in enumerate(names):
        def launch(v=v):
            rc = S.lib.smh_stream_read_probe_variant(C.c_void_p(R.text.data_ptr()), R.per_gpu, C.c_void_p(probe.data_ptr()), C.c_void_p(R.stream), v)
            if rc != 0:
                raise SystemExit("stream probe %d: %s" % (v, S.lib.smh_last_error().decode()))
        pms = sorted(R.timed(launch, 6, probe))[2]
        variants[nm] = dict(ms=round(pms, 4), GBps=R.rate(R.per_gpu, pms)["GBps"])
    best = min(variants, key=lambda k: variants[k]["ms"])
    out["stream_read"] = dict(kernel="smh_stream_read_probe_variant: best of %d read-only kernels (no table work)" % len(names),
                              best=best, ms=variants[best]["ms"],
                              **{k: v for k, v in R.rate(R.per_gpu, variants[best]["ms"]).items() if k != "Gbit_s"}, variants=variants)
    out["roofline"]["of_stream_read"] = round(out["roofline"]["achieved"] / out["stream_read"]["GBps"], 4)

    m_pos = 16
    cap = max(1024, 2 * int(R.local_counts[AC_LENGTHS.index(m_pos)]))
    pbuf = torch.zeros(cap, dtype=torch.int64, device=R.dev)
    pcur = torch.zeros(1, dtype=torch.int64, device=R.dev)
    pms = sorted(R.timed(lambda: R.acs[m_pos].positions_device(R.text.data_ptr(), R.shard_len[m_pos], pbuf.data_ptr(), cap,
                                                               pcur.data_ptr(), R.stream), 6, pcur))[2]
    out["positions"] = dict(workload="smh_ac_positions, m=%d set, same text: END columns of all matches" % m_pos,
                            kernel_ms=round(pms, 4), GBps=R.rate(R.shard_len[m_pos], pms)["GBps"], matches=int(pcur.item()),
                            equals_count=int(pcur.item()) == R.local_counts[AC_LENGTHS.index(m_pos)])


def phase_wm_and_automaton(R):
    """BASELINE configs[2] (WM, same text, 10 000 x 8), the headline's longer sets through the Wu-Manber entry point, and the
    headline sets whose entry point chose the pair-gram filter forced onto the automaton kernels (`roofline.automaton`)"""
    args, S, sharded, out = R.args, R.S, R.sharded, R.out
    R.wpat = S.corpus_patterns(WM_LENGTH, WM_PATTERNS, PAT_SEED + 1, SIGMA, TEXT_SEED, R.n_total, 2)
    wb, we = sharded.shard_for_rank(R.n_total, R.world, R.rank, WM_LENGTH)
    t0 = time.perf_counter()
    obj, R.wm = R.sharded_set("wm", "wm", R.wpat, WM_LENGTH, WM_PATTERNS, SIGMA, R.text, we - wb, args.steps)
    if R.rank == 0:
        wi = R.wm.info()
        out["wm"] = dict(workload="WM: same text, %d patterns of length %d (BASELINE configs[2]); rate = all ranks' bytes / "
                                  "slowest device's kernel time" % (WM_PATTERNS, WM_LENGTH), **obj,
                         block_symbols=wi.block_symbols, filter_log2=wi.filter_log2, filter_exact=wi.filter_exact,
                         shift_zero="%d/%d" % (wi.shift_zero, wi.shiftsize))
    # the headline's longer pattern sets (m = 16, 32; the same 1000 patterns) through the Wu-Manber entry point
    wl = {}
    for m in AC_LENGTHS[1:]:
        obj, wml = R.sharded_set("wm_long.m%d" % m, "wm", R.pats[m], m, AC_PATTERNS, SIGMA, R.text, R.shard_len[m], args.steps)
        li = wml.info()
        wl["m%d" % m] = dict(**obj, scan_engine="automaton kernels" if li.scan_engine == S.ALGO_AC else "suffix-filter kernels",
                             gram_planes=li.gram_planes)
    if R.rank == 0:
        out["wm_long"] = dict(workload="WM: same text, the headline's %d-pattern sets of length %s through the Wu-Manber entry "
                                       "point (q-gram shift-or filter in LDS + staged verify)" % (AC_PATTERNS, "/".join(str(m) for m in AC_LENGTHS[1:])), **wl)

    # the headline sets whose depth-cut automaton plan handed the scan to the pair-gram filter (ac_host.c, end of the
    # compile; `ac.mNN.scan_engine`), through the AUTOMATON kernels all the same: what the choice is worth, and parity
    # of the engine that is not the default
    aa = {}
    for m in AC_LENGTHS:
        if R.acs[m].info().scan_engine == S.ALGO_WM:
            obj, h = R.sharded_set("ac_automaton.m%d" % m, "ac", R.pats[m], m, AC_PATTERNS, SIGMA, R.text, R.shard_len[m], args.steps,
                                   engine=S.ALGO_AC)
            hi = h.info()
            aa["m%d" % m] = dict(**obj, scan_stride=hi.scan_stride, scan_depth=hi.scan_depth, scan_exact=hi.scan_exact,
                                 scan_full_rows=hi.scan_full_rows, lds_bytes=hi.lds_bytes, kernel_instance=ac_kernel_name(hi))
    if R.rank != 0:
        return
    # the line's roofline names the kernel the entry point chose for the slowest headline set; beside it, the automaton
    # kernels (ac_dfa_kernel) on the same set: the engine north_star describes, whichever one the entry point runs
    dkey = "m%d" % R.dom
    if dkey in aa:
        out["roofline"]["automaton"] = dict(kernel_instance=aa[dkey]["kernel_instance"], launch_ms=aa[dkey]["kernel_ms"],
                                            achieved=aa[dkey]["GBps"], frac=aa[dkey]["hbm_frac"],
                                            what="the same set with smh_ac_set_scan_engine(SMH_ALGO_AC): the automaton kernels")
    else:
        out["roofline"]["automaton"] = dict(kernel_instance=out["roofline"]["kernel_instance"], launch_ms=out["roofline"]["launch_ms"],
                                            achieved=out["roofline"]["achieved"], frac=out["roofline"]["frac"],
                                            what="the entry point runs the automaton kernels for this set")
    out["roofline"]["chosen_engine"] = out["ac"][dkey]["scan_engine"]
    if aa:
        out["ac_automaton"] = dict(workload="AC: the headline sets whose entry point chose the pair-gram filter, forced onto the "
                                            "automaton kernels (smh_ac_set_scan_engine(SMH_ALGO_AC)): hybrid stride-2 image, depth-cut, "
                                            "three chains per lane", **aa)


def phase_mixed_lengths(R):
    """N = 1: BASELINE configs[1] read literally -- ONE set of 1000 patterns whose lengths run from 8 to 32 (40 per length) --
    through the pattern-set entry points (smh_pset_*: the reference API carries one length per run)"""
    if R.rank != 0 or R.world != 1:
        return
    S, torch, np = R.S, R.torch, R.np
    mlens, mpats = [], []
    for L in range(8, 33):
        mpats.append(S.corpus_patterns(L, 40, PAT_SEED + 100 + L, SIGMA, TEXT_SEED, R.n_total, 2))
        mlens += [L] * 40
    R.mixed = (np.concatenate(mpats), np.array(mlens, dtype=np.uint32))
    mcount = torch.zeros(1, dtype=torch.int64, device=R.dev)
    mobj = {}
    for name, algo in (("ac", S.ALGO_AC), ("wm", S.ALGO_WM)):
        ps = S.PatternSet(R.mixed[0], R.mixed[1], SIGMA, algo)
        mls = R.timed(lambda: ps.scan_device(R.text.data_ptr(), R.per_gpu, mcount.data_ptr(), R.stream), R.args.steps, mcount)
        ms = sum(mls) / len(mls)
        mobj[name] = dict(kernel_ms=round(ms, 4), min_ms=round(min(mls), 4), **R.rate(R.per_gpu, ms), matches=int(mcount.item()),
                          one_pass=int(ps.info().one_pass), classes=int(ps.info().classes))
        ps.close()
    R.out["mixed_8_32"] = dict(workload="BASELINE configs[1] read as ONE set: 1000 patterns, 40 of each length 8..32, same text, "
                                        "scanned in one pass; count = sum over length classes of the reference's count", **mobj)


def phase_shard_configs(R):
    """The 32 GB configurations: every rank scans ITS 4 GiB byte range (+ halo) of one (N x 4 GiB) text"""
    args, S, torch, sharded, out = R.args, R.S, R.torch, R.sharded, R.out

    def shard_config(label, algo, sigma, lengths, p, seed, workload):
        n_tot = R.shard * R.world
        b0, resident, lens = sharded.shard_plan(n_tot, R.world, R.rank, lengths)
        reuse = sigma == SIGMA and R.world == 1 and R.shard == R.per_gpu
        t = R.text if reuse else R.corpus(resident, b0, sigma)
        objs = {}
        for m in lengths:
            pat = S.corpus_patterns(m, p, seed, sigma, TEXT_SEED, n_tot, 2)
            obj, h = R.sharded_set("%s.m%d" % (label, m), algo, pat, m, p, sigma, t, lens[m], 5)
            if algo == "ac":
                i4 = h.info()
                obj.update(kernel_instance=ac_kernel_name(i4),
                           scan_engine="suffix-filter kernels" if i4.scan_engine == S.ALGO_WM else "automaton kernels",
                           scan_stride=i4.scan_stride, scan_depth=i4.scan_depth)
            objs["m%d" % m] = obj
        if R.rank == 0:
            out[label] = dict(workload=workload, text_bytes_total=n_tot, sharding="byte-range x%d, m-1 halo, counts all-reduced" % R.world, **objs)
            if algo == "ac" and "stream_read" in out and not reuse:
                # the streaming-read ceiling for a shard of THIS size (a 4 GiB launch amortises its start and end better
                # than a 1 GiB one: the probes read 1-2 % faster on it, and so do the scan kernels)
                probe = torch.zeros(1, dtype=torch.int64, device=R.dev)
                best = None
                for v in range(5):
                    def launch(v=v):
                        rc = S.lib.smh_stream_read_probe_variant(C.c_void_p(t.data_ptr()), R.shard, C.c_void_p(probe.data_ptr()), C.c_void_p(R.stream), v)
                        if rc != 0:
                            raise SystemExit("stream probe %d: %s" % (v, S.lib.smh_last_error().decode()))
                    pms = sorted(R.timed(launch, 4, probe))[1]
                    best = pms if best is None or pms < best else best
                out["stream_read"]["shard"] = dict(bytes=R.shard, ms=round(best, 4), **{k: v for k, v in R.rate(R.shard, best).items() if k != "Gbit_s"})

    # BASELINE configs[3]: AC, 8000 patterns; 32 GB over 8 GPUs = a 4 GiB byte range per GPU
    shard_config("ac_8000_patterns", "ac", SIGMA, AC_LENGTHS, C4_PATTERNS, PAT_SEED + 3,
                 "AC: %d MiB of DNA text per GPU (BASELINE configs[3]: 32 GB over 8 GPUs), 8000 patterns per set, m=8/16/32; "
                 "scan_engine says which kernels served the Aho-Corasick entry point" % args.shard_mib)
    # BASELINE configs[4]: WM, 256-symbol alphabet, 100 000 patterns, lengths 5-20 as fixed-length sets
    shard_config("wm_ascii", "wm", C5_SIGMA, C5_LENGTHS, C5_PATTERNS, PAT_SEED + 2,
                 "WM: %d MiB of 256-symbol text per GPU (BASELINE configs[4]), 100000 patterns per set, m=%s"
                 % (args.shard_mib, "/".join(str(m) for m in C5_LENGTHS)))
    # the lengths at the byte-gram forms' boundaries (flat <= 9 / hashed above), same shard: kernel time + slice verification
    shard_config("wm_ascii_more", "wm", C5_SIGMA, C5_MORE_LENGTHS, C5_PATTERNS, PAT_SEED + 2,
                 "WM: the remaining lengths of BASELINE configs[4]'s 5-20 sweep that sit at filter-form boundaries, m=%s; verified on "
                 "the first 512 MiB + the last 64 MiB of the shard" % "/".join(str(m) for m in C5_MORE_LENGTHS))


def phase_skewed(R):
    """N = 1: text that is NOT i.i.d. uniform (round 4) -- the BASELINE pattern shapes on genome-like / protein-like / natural-
    language-like text and on a text in which one pattern recurs every 64 columns, patterns sampled from those texts"""
    if R.rank != 0 or R.world != 1:
        return
    args, S, torch = R.args, R.S, R.torch
    per_gpu, stream = R.per_gpu, R.stream
    sk = {}
    corpora = [("dna_repeats", S.CORPUS_DNA_REPEATS, 4), ("dna_planted", S.CORPUS_PLANTED, 4),
               ("protein_skewed", S.CORPUS_SKEWED, 20), ("ascii_skewed", S.CORPUS_SKEWED, 256),
               ("protein_uniform", S.CORPUS_UNIFORM, 20)]  # the 20-letter alphabet on uniform text: no BASELINE configuration covers it
    shapes = {4: [("ac", AC_PATTERNS, 8), ("ac", AC_PATTERNS, 16), ("ac", AC_PATTERNS, 32), ("wm", WM_PATTERNS, WM_LENGTH),
                  ("ac", C4_PATTERNS, 16), ("ac", C4_PATTERNS, 32)],
              20: [("ac", AC_PATTERNS, 8), ("ac", AC_PATTERNS, 16), ("wm", WM_PATTERNS, WM_LENGTH)],
              256: [("wm", C5_PATTERNS, 8), ("wm", C5_PATTERNS, 12), ("wm", C5_PATTERNS, 20)]}
    scnt = torch.zeros(1, dtype=torch.int64, device=R.dev)
    worst_ratio = 0.0
    for cname, kind, sigma in corpora:
        ktext = torch.empty(per_gpu + 64, dtype=torch.uint8, device=R.dev)
        S.corpus_text_device(ktext.data_ptr(), per_gpu, TEXT_SEED, sigma, 0, kind, stream)
        torch.cuda.synchronize()
        cobj = {}
        for algo, p, m in shapes[sigma]:
            pat = S.corpus_patterns(m, p, PAT_SEED + 5, sigma, TEXT_SEED, per_gpu, 2, kind)
            h = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
            launch = lambda: h.scan_device(ktext.data_ptr(), per_gpu, scnt.data_ptr(), S.VARIANT_TUNED, stream)
            # adaptation: launches with a synchronisation behind each, so that each one's report is read before the next
            seq, first_ms = [], None
            for it in range(6):
                scnt.zero_()
                e0, e1 = R.ev(), R.ev()
                e0.record()
                launch()
                e1.record()
                torch.cuda.synchronize()
                if it == 0:  # a fresh handle's first call on this text: table upload aside, the first look
                    first_ms = e0.elapsed_time(e1)
                seq.append(int(h.adapt().engine))
            settled = next((i for i in range(len(seq)) if all(e == seq[-1] for e in seq[i:])), len(seq))
            ms = sorted(R.timed(launch, 5, scnt))[2]
            matches = int(scnt.item())
            ad = h.adapt()
            rec = dict(patterns=p, m=m, entry=algo, adaptive=int(h.info().adaptive), compiled_engine=S.ENGINE_NAMES[int(h.info().scan_engine)],
                       chosen=dict(engine=S.ENGINE_NAMES[int(ad.engine)], kernel_ms=round(ms, 4), **R.rate(per_gpu, ms), flips=int(ad.flips),
                                   engines_per_launch=seq, launches_before_settled=settled, first_launch_ms=round(first_ms, 3),
                                   events_per_4k=round(ad.events_per_4k[int(ad.engine)], 3)),
                       matches=matches)
            forced, equal = {}, True
            for eng in (S.ALGO_AC, S.ALGO_WM, S.ENGINE_AC_FLAT, S.ENGINE_KEYS, S.ENGINE_HASH):
                try:
                    h.set_scan_engine(eng)
                except S.SmhError:
                    continue
                fms = sorted(R.timed(launch, 3, scnt))[1]
                forced[S.ENGINE_NAMES[eng]] = dict(kernel_ms=round(fms, 4), hbm_frac=R.rate(per_gpu, fms)["hbm_frac"], matches=int(scnt.item()))
                equal = equal and int(scnt.item()) == matches
            h.set_scan_engine(-1)
            rec["forced"] = forced
            rec["engines_agree"] = equal
            rec["key_slots"] = int(h.info().key_slots)  # > 0: the handle keeps the key engine (round 5)
            if algo == "wm":
                rec["hash_slots"] = int(h.info().hash_slots)  # > 0: ... the window-hash engine
            if algo == "ac":
                rec["flat_parts"] = int(h.info().flat_parts)  # launches of the text-independent engine
            every = [v["kernel_ms"] for v in forced.values()]  # all the handle holds, the text-independent parts included
            if every:
                rec["chosen_vs_best_forced"] = round(ms / min(every), 3)
                worst_ratio = max(worst_ratio, ms / min(every))
            if not equal:
                R.out["skewed"] = {cname: {"%s_%d_m%d" % (algo, p, m): rec}}
                fail_run(R, "engines disagree on %s %s p=%d m=%d" % (cname, algo, p, m))
            R.verify.append(("skewed.%s.%s_%d_m%d" % (cname, algo, p, m), "ac" if sigma == 4 else "wm", pat, m, p, sigma, ktext, per_gpu, matches, None))
            cobj["%s_%d_m%d" % (algo, p, m)] = rec
            h.close()
        sk[cname] = cobj
    R.out["skewed"] = dict(workload="the BASELINE pattern shapes on %d MiB of non-uniform text per corpus (csrc/corpus_gen.h; protein_uniform: the 20-letter "
                                    "alphabet on uniform text), patterns sampled from the text; chosen = the entry point as compiled after 6 launches (the adaptive engine follows the launches' "
                                    "reports; first_launch_ms = the fresh handle's first scan of this text, device time: over 1 GiB or more it looks at the first 256 MiB "
                                    "with the compile's choice and, when that runs 3x over its estimate, with the other engines, before the rest is launched), "
                                    "forced = smh_*_set_scan_engine with every engine the handle holds (the text-independent one = the set as "
                                    "flat_parts exact stride-1 automata scanned one after the other); chosen_vs_best_forced = chosen / the fastest forced" % args.mib_per_gpu,
                           worst_chosen_vs_best_forced=round(worst_ratio, 3), **sk)


def phase_table_kernels(R):
    """N = 1: the table-walking kernels (cuda_*1/2: the reference's tables walked as given) on a 64 MiB prefix"""
    if R.rank != 0 or R.world != 1:
        return
    S, torch, text, stream = R.S, R.torch, R.text, R.stream
    tn = min(64 << 20, R.per_gpu)
    tk, tcnt = {}, torch.zeros(1, dtype=torch.int64, device=R.dev)
    tpat = R.pats[8]
    sets = [("ac_table_kernel (cuda_ac1/2)", R.acs[8], lambda h: h.scan_device(text.data_ptr(), tn, tcnt.data_ptr(), S.VARIANT_TABLE, stream)),
            ("wm_table_kernel (cuda_wm1/2)", S.WmTables.from_patterns(tpat, 8, AC_PATTERNS, SIGMA),
             lambda h: h.scan_device(text.data_ptr(), tn, tcnt.data_ptr(), S.VARIANT_TABLE, stream)),
            ("sh_table_kernel (cuda_sh1/2)", S.ShTrie.from_patterns(tpat, 8, AC_PATTERNS, SIGMA),
             lambda h: h.scan_device(text.data_ptr(), tn, tcnt.data_ptr(), None, S.VARIANT_TABLE, stream)),
            ("sbom_table_kernel (cuda_sbom1/2)", S.SbomOracle.from_patterns(tpat, 8, AC_PATTERNS, SIGMA),
             lambda h: h.scan_device(text.data_ptr(), tn, tcnt.data_ptr(), S.VARIANT_TABLE, stream)),
            ("sog_table_kernel (cuda_sog1/2)", S.SogTables(tpat, AC_PATTERNS),
             lambda h: h.scan_device(text.data_ptr(), tn, tcnt.data_ptr(), S.VARIANT_TABLE, stream))]
    want = None
    for nm, h, launch in sets:
        tms = sorted(R.timed(lambda: launch(h), 3, tcnt))[1]
        tk[nm] = dict(kernel_ms=round(tms, 4), **R.rate(tn, tms), matches=int(tcnt.item()))
        want = int(tcnt.item()) if want is None else want
        if int(tcnt.item()) != want:
            fail_run(R, "%s counted %d, ac_table_kernel %d" % (nm, int(tcnt.item()), want))
    R.verify.append(("table_kernels", "ac", tpat, 8, AC_PATTERNS, SIGMA, text, tn, want, None))
    R.out["table_kernels"] = dict(workload="the reference-layout tables walked as given (latency-bound by design), m=8 set of %d "
                                           "patterns, first %d MiB of the same text; all five counts equal" % (AC_PATTERNS, tn >> 20), **tk)


# the reference's own data sets (main.c:39-109; the files are not in its repository): bytes and alphabet
SMALL_TEXTS = [("world192", 1903104, 128), ("random", 3999744, 8), ("E.coli", 4628736, 4), ("A.thaliana.faa", 10821888, 20),
               ("A.thaliana.fna", 116234496, 4), ("swiss-prot", 177649920, 20)]
SMALL_SETS = [("ac", 1000, 8), ("wm", 8000, 8)]  # execute.sh:8-9,16-51: m = 8, p_size in {1000, 8000}


def phase_small_text(R):
    """N = 1 (round 6): texts of the sizes the reference itself benchmarks (main.c:39-109: 1.9 MB .. 178 MB) -- launches of tens of
    microseconds, where what a launch costs before and after its streaming part decides the rate.  Per size and set: the launch's
    device time (events around 20 back-to-back launches / 20), the same launches replayed from a captured hipGraph, and a
    breakdown: floor = the same handle over ONE wave-chunk (dispatch + the table staged into every workgroup's LDS + the final
    reduction), stream = bytes / the handle's 1 GiB rate, rest = what is left (start skew, tail)."""
    if R.rank != 0 or R.world != 1:
        return
    S, torch, stream = R.S, R.torch, R.stream
    reps = 20
    cnt = torch.zeros(1, dtype=torch.int64, device=R.dev)
    obj = {}
    big_n = min(R.per_gpu, 1 << 30)
    texts = {SIGMA: R.text}  # one text of big_n bytes per alphabet: the data-set sizes are prefixes of it, the 1 GiB reference its whole
    for name, n, sigma in SMALL_TEXTS:
        if sigma not in texts:
            texts[sigma] = R.corpus(big_n, 0, sigma)
            torch.cuda.synchronize()
        t = texts[sigma]
        rec = {}
        for algo, p, m in SMALL_SETS:
            pat = S.corpus_patterns(m, p, PAT_SEED + 7, sigma, TEXT_SEED, n, 2)
            h = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
            launch = lambda nn=n: h.scan_device(t.data_ptr(), nn, cnt.data_ptr(), S.VARIANT_TUNED, stream)

            def burst_ms(fn, k=reps):
                """device time per launch of k launches issued back to back (events around the burst)"""
                best = None
                for _ in range(5):
                    a, b = R.ev(), R.ev()
                    a.record()
                    for _ in range(k):
                        fn()
                    b.record()
                    torch.cuda.synchronize()
                    ms = a.elapsed_time(b) / k
                    best = ms if best is None or ms < best else best
                return best
            launch()
            torch.cuda.synchronize()
            R.conditioned(launch, 0.05, ms=5.0)
            cnt.zero_()
            launch()
            torch.cuda.synchronize()
            matches = int(cnt.item())
            ms = burst_ms(launch)
            floor = burst_ms(lambda: h.scan_device(t.data_ptr(), 4096, cnt.data_ptr(), S.VARIANT_TUNED, stream))
            if "empty_launch_ms" not in obj:  # a kernel without a table to stage over the same 4 KiB: what of the floor is the dispatch itself
                probe = torch.zeros(1, dtype=torch.int64, device=R.dev)
                obj["empty_launch_ms"] = round(burst_ms(lambda: S.lib.smh_stream_read_probe(C.c_void_p(t.data_ptr()), 4096, C.c_void_p(probe.data_ptr()), C.c_void_p(stream))), 5)
            # the same 20 launches as ONE captured graph: what a caller that replays a graph pays per launch
            graph_ms = None
            try:
                side = torch.cuda.Stream(device=R.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, side.cuda_stream)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(reps):
                            h.scan_device(t.data_ptr(), n, cnt.data_ptr(), S.VARIANT_TUNED, side.cuda_stream)
                    best = None
                    for _ in range(5):
                        a, b = R.ev(), R.ev()
                        a.record(side)
                        g.replay()
                        b.record(side)
                        torch.cuda.synchronize()
                        gm = a.elapsed_time(b) / reps
                        best = gm if best is None or gm < best else best
                    graph_ms = best
            except Exception as e:  # noqa: BLE001  (a capture that fails costs this figure, not the run)
                graph_ms = None
                rec.setdefault("graph_errors", []).append(repr(e)[:160])
            big = sorted(R.timed(lambda: h.scan_device(t.data_ptr(), big_n, cnt.data_ptr(), S.VARIANT_TUNED, stream), 3, cnt))[1]
            stream_ms = n / big_n * big
            r = dict(kernel_ms=round(ms, 5), **R.rate(n, ms), matches=matches, floor_ms=round(floor, 5),
                     graph_replay_ms=round(graph_ms, 5) if graph_ms else None,
                     graph_hbm_frac=R.rate(n, graph_ms)["hbm_frac"] if graph_ms else None)
            r.update(gib_kernel_ms=round(big, 4), gib_hbm_frac=R.rate(big_n, big)["hbm_frac"], stream_ms=round(stream_ms, 5),
                     rest_ms=round(ms - floor - stream_ms, 5), of_gib_rate=round((n / ms) / (big_n / big), 3))
            rec["%s_%d_m%d" % (algo, p, m)] = r
            R.verify.append(("small_text.%s.%s_%d_m%d" % (name, algo, p, m), algo, pat, m, p, sigma, t, n, matches, None))
            h.close()
        obj[name] = dict(bytes=n, alphabet=sigma, **rec)
    empty = obj.pop("empty_launch_ms", None)
    R.out["small_text"] = dict(workload="uniform synthetic text of the reference's data-set sizes and alphabets (main.c:39-109), AC 1000 x 8 and WM 8000 x 8 "
                                        "(execute.sh:8-9); kernel_ms = device time per launch of %d back-to-back launches; floor_ms = the same over one "
                                        "4 KiB wave-chunk; stream_ms = bytes x the handle's 1 GiB time; rest_ms = kernel - floor - stream; of_gib_rate = this "
                                        "size's rate / the 1 GiB rate; graph_replay_ms = per launch of the same %d launches replayed from one captured graph; "
                                        "empty_launch_ms = the streaming-read probe (no table) over 4 KiB, same protocol" % (reps, reps),
                               empty_launch_ms=empty, **obj)


def phase_preproc(R):
    """rank 0 (round 6): what the reference times as `preproc` beside `search` (main.c:246-262, 277-293): the reference-shaped
    fill of the caller's tables (preproc_ac / preproc_wu2, host), the handle compile, and a fresh handle's first scan of a short
    text (its table set goes up, its code object loads) -- per BASELINE set.  configs[4]'s dense PREFIX arrays would be 2 x 2.1 GB
    (main.c:436-439): its legacy fill is not run, the handle is compiled from the patterns as bench.py does everywhere."""
    if R.rank != 0:
        return
    S, torch, np = R.S, R.torch, R.np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cnt = torch.zeros(1, dtype=torch.int64, device=R.dev)
    short = {}
    rows = {}

    def one(label, algo, sigma, p, m, seed, legacy=True):
        pat = S.corpus_patterns(m, p, seed, sigma, TEXT_SEED, R.n_total, 2)
        rec = {}
        if legacy and algo == "ac":
            rec["preproc_ac_s"] = round(time_preproc_ac(S, np, pat, m, p, sigma), 5)
        elif legacy:
            rec["preproc_wu2_s"] = round(time_preproc_wu2(S, np, pat, m, p, sigma)[0], 5)
        t0 = time.perf_counter()
        h = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
        rec["compile_s"] = round(time.perf_counter() - t0, 5)
        if sigma not in short:
            short[sigma] = R.corpus(1 << 20, 0, sigma)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.scan_device(short[sigma].data_ptr(), 1 << 20, cnt.data_ptr(), S.VARIANT_TUNED, R.stream)
        torch.cuda.synchronize()
        rec["first_scan_s"] = round(time.perf_counter() - t0, 5)
        rec["preproc_s"] = round(sum(v for k, v in rec.items() if k.endswith("_s")), 5)
        h.close()
        rows[label] = rec

    for m in AC_LENGTHS:
        one("ac_%d_m%d" % (AC_PATTERNS, m), "ac", SIGMA, AC_PATTERNS, m, PAT_SEED)
    if not R.args.no_wm:
        one("wm_%d_m%d" % (WM_PATTERNS, WM_LENGTH), "wm", SIGMA, WM_PATTERNS, WM_LENGTH, PAT_SEED + 1)
        for m in AC_LENGTHS:
            one("ac_%d_m%d" % (C4_PATTERNS, m), "ac", SIGMA, C4_PATTERNS, m, PAT_SEED + 3)
        for m in C5_LENGTHS:
            one("wm_ascii_%d_m%d" % (C5_PATTERNS, m), "wm", C5_SIGMA, C5_PATTERNS, m, PAT_SEED + 2, legacy=False)
    # the legacy GPU names keep the handle they compiled (smh_runtime.hip): main.c:623-648's five calls, one build
    builds0 = int(S.lib.smh_legacy_handle_builds())
    calls_s = time_cuda_wm_calls(S, np, R.wpat if R.wpat is not None else S.corpus_patterns(WM_LENGTH, WM_PATTERNS, PAT_SEED + 1, SIGMA, TEXT_SEED, R.n_total, 2),
                                 WM_LENGTH, WM_PATTERNS, SIGMA, R.text[:4 << 20].cpu().numpy())
    S.lib.smh_host_path_release()
    R.out["preproc"] = dict(what="seconds, host wall clock: preproc_* = the reference-shaped fill of caller-owned tables (include/smatcher.h; O(states) "
                                 "breadth-first pass where ac/list.h:57-74 walks the queue per append), compile_s = patterns -> handle (every engine the "
                                 "handle keeps, LDS images, cost model), first_scan_s = the fresh handle's first scan of 1 MiB (table set to the device, "
                                 "code object load); preproc_s = their sum.  The headline's own: compile %s s, first three scans together %.4f s"
                                 % ("/".join("%.4f" % R.compile_s[m] for m in AC_LENGTHS), R.first_scans_s),
                            sets=rows,
                            legacy_cuda_wm_calls=dict(what="cuda_wm1..5 back to back on one set of caller tables (main.c:623-648), 4 MiB of text: seconds per call and "
                                                           "handles compiled -- the first call compiles and uploads, the other four reuse its handle (cuda_wm1 / 2 walk the reference tables as given: slow by design, see table_kernels)",
                                                      seconds=[round(x, 5) for x in calls_s], handle_builds=int(S.lib.smh_legacy_handle_builds()) - builds0))


def phase_multi_leg(R):
    """The same workloads through the native one-process path (smh_multi_*): a CHILD process of rank 0 drives all N devices
    while the ranks wait in the c10d store with their GPUs idle"""
    args, out = R.args, R.out
    R.torch.cuda.synchronize()
    R.sharded.host_barrier("smh_multi_before")
    if R.rank == 0:
        cmd = [sys.executable, os.path.abspath(__file__), "--multi-leg", str(R.world), "--steps", "5", "--mib-per-gpu", str(args.mib_per_gpu),
               "--shard-mib", str(args.shard_mib)] + (["--no-wm"] if args.no_wm else [])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            leg = json.loads(line[-1]) if r.returncode == 0 and line else {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            leg = {"error": repr(e)[:400]}
        # same text, same patterns, same byte ranges: the totals must be the per-rank path's
        agree = {}
        for key, mine in (("ac", out.get("ac")), ("ac_8000_patterns", out.get("ac_8000_patterns")), ("wm_ascii", out.get("wm_ascii"))):
            for mk, v in (leg.get(key) or {}).items():
                if mk.startswith("m") and mine and mk in mine:
                    agree["%s.%s" % (key, mk)] = v["matches"] == mine[mk]["matches"]
        leg["totals_equal_per_rank_path"] = agree
        out["smh_multi"] = leg
        R.multi_ok = bool(agree) and all(agree.values()) and "error" not in leg
        if agree and not all(agree.values()):
            fail_run(R, "smh_multi totals differ from the per-rank totals: %r" % agree)
        if "error" in leg or not agree:
            # a leg that did not run is not a leg that agreed: `smh_multi_ok` on the compact line is false and the run ends non-zero
            # AFTER the record is out -- the per-rank measurement above is complete and verified by itself (until round 5 a failed
            # side leg took the headline with it; in round 5 it vanished into an optional group of the line)
            leg.setdefault("error", "no totals to compare")
            print("bench.py: smh_multi leg did not run: %s" % leg["error"], file=sys.stderr)
    R.sharded.host_barrier("smh_multi_after")


def phase_verify(R):
    """bit-exact verification of every count above, every rank its own shards"""
    sharded, out = R.sharded, R.out
    cpu = R.cpu = Cpu(R.world)
    t0 = time.perf_counter()
    mine, host_cache = {}, {}
    longest = {}  # (text, offset) -> the longest slice any entry wants from there: copied once, sliced per entry

    def host_slice(dtext, off, ln):
        key = (id(dtext), off)
        if key not in host_cache or len(host_cache[key]) < ln:
            host_cache.clear()  # one (up to 4 GiB) host copy at a time
            host_cache[key] = dtext[off:off + max(ln, longest.get(key, 0))].cpu().numpy()
        return host_cache[key][:ln]
    R.host_slice = host_slice

    def slices_of(n, m, name=""):
        # the headline's 1 GiB shards are recounted whole at every N; the 4 GiB shards within the budget
        if name.startswith("wm_ascii_more"):
            return sharded.verify_slices(n, m, 576 << 20)
        return sharded.verify_slices(n, m, 0 if n <= R.per_gpu + 64 else R.verify_budget)

    verify = R.verify
    for v in verify:
        for off, ln in slices_of(v[7], v[3], v[0]):
            longest[(id(v[6]), off)] = max(longest.get((id(v[6]), off), 0), ln)
    verify.sort(key=lambda v: id(v[6]))  # shards of the same text together: fewer device-to-host copies
    recounted = {}  # the same set over the same bytes by the same checker (ac.m16 / ac_automaton.m16: one set, two engines) is recounted once
    for name, algo, pat, m, p, sigma, dtext, n, got, scan in verify:
        slices = slices_of(n, m, name)
        count = None
        g, c = [], []
        for off, ln in slices:
            g.append(got if (off == 0 and ln == n) or scan is None else scan(dtext.data_ptr() + off, ln))
            key = (algo, id(pat), m, p, sigma, id(dtext), off, ln)
            if key not in recounted:
                count = count or cpu.counter(algo, pat, m, p, sigma)
                recounted[key] = count(host_slice(dtext, off, ln))
            c.append(recounted[key])
        mine[name] = dict(gpu=g, cpu=c, slices=[[o, l] for o, l in slices], shard_bytes=n)
    if R.mixed is not None:  # the mixed-length set: sum over its 25 length classes of the restated search_ac, full text
        want, host_text = 0, host_slice(R.text, 0, R.per_gpu)
        for L in range(8, 33):
            flat = R.mixed[0][sum(R.mixed[1][:(L - 8) * 40]):sum(R.mixed[1][:(L - 8) * 40]) + 40 * L]
            want += cpu.counter("ac", flat, L, 40, SIGMA)(host_text)
        for name in ("ac", "wm"):
            mine["mixed_8_32." + name] = dict(gpu=[out["mixed_8_32"][name]["matches"]], cpu=[int(want)], slices=[[0, R.per_gpu]], shard_bytes=R.per_gpu)
    my_secs = time.perf_counter() - t0
    merged, all_equal = sharded.merge_verified(sharded.gather_objects(mine))
    secs = max(sharded.gather_objects(my_secs))
    R.parity_ok = R.parity_ok and all_equal
    if R.rank == 0:
        out["verified"] = dict(checker="restated search_ac / search_wu2 (oracle/, pinned to the reference on the golden vectors) over "
                                       "byte-range pieces with an m-1 halo, every rank its own shards on %d of the host's %d threads "
                                       "(%d CPUs); %s" % (cpu.cores, cpu.host_threads, cpu.host_cpus,
                                                          "full text of every configuration" if R.verify_budget == 0 else
                                                          "1 GiB shards whole, 4 GiB shards: first %d MiB + last 64 MiB (with the halo)"
                                                          % ((R.verify_budget >> 20) - 64)),
                               seconds=round(secs, 1), all_equal=all_equal, counts=merged)


def phase_cpu_baselines(R):
    """N = 1, rank 0: the reference's own compiled search_ac / search_wu2 (oracle/_ref; the restated port when absent) on bounded
    prefixes, one thread and all threads; the legacy host-pointer path"""
    args, S, np, out, cpu = R.args, R.S, R.np, R.out, R.cpu
    per_gpu, text, acs, pats = R.per_gpu, R.text, R.acs, R.pats
    sample = min(args.cpu_sample_mib << 20, per_gpu)
    host_text = text[:per_gpu].cpu().numpy()
    prefix = S.corpus_text(4096, TEXT_SEED, SIGMA, offset=0)
    assert np.array_equal(prefix, host_text[:4096]), "device and host corpus generators differ"
    prefix = host_text[:sample]
    # serial search_ac on the prefix
    secs, cpu_counts = cpu.ac_serial(pats, AC_PATTERNS, SIGMA, prefix)
    what = ("the reference's own ac/ac.c compiled where it lies (oracle/_ref/libref.so)" if cpu.kind == "reference"
            else "the restated port oracle/ora_ac.c (the reference's sources were not there to compile)")
    out["cpu_baseline"] = dict(value=round(8.0 * sample * len(pats) / secs / 1e9, 4), unit="Gbit/s", cores=1, kind=cpu.kind,
                               cpu=cpu.model, host_cpus=cpu.host_cpus,
                               sample="search_ac (ac/ac.c:198-222; %s) over the first %d MiB of the same text, m=%s, %d patterns "
                                      "each, 1 thread, %.1f s" % (what, sample >> 20, "/".join(str(m) for m in pats), AC_PATTERNS, secs))
    gpu_counts = {m: verify_scan(text.data_ptr(), sample) for m, verify_scan in ((m, R.scan_with(acs[m])) for m in AC_LENGTHS)}
    ok = all(gpu_counts[m] == cpu_counts[m] for m in AC_LENGTHS)
    out["parity"] = dict(bit_exact=ok, gpu_counts=gpu_counts, cpu_counts=cpu_counts, sample_bytes=sample)
    R.parity_ok = R.parity_ok and ok
    # serial search_wu2 on its (smaller) prefix: the 3-symbol SHIFT table is all zero on DNA from ~1000 patterns up,
    # so every column scans a bucket (0.036 Gbit/s on one thread)
    wsample = wcnt = None
    if R.wpat is not None:
        wsample = min(args.cpu_wm_sample_mib << 20, per_gpu)
        wsecs, wcnt = cpu.wm_serial(R.wpat, WM_LENGTH, WM_PATTERNS, SIGMA, host_text[:wsample])
        out["cpu_baseline_wm"] = dict(value=round(8.0 * wsample / wsecs / 1e9, 4), unit="Gbit/s", cores=1, kind=cpu.kind,
                                      cpu=cpu.model, counts_match=R.scan_with(R.wm)(text.data_ptr(), wsample) == wcnt,
                                      sample="search_wu2 (wu/wu.c:151-209) over the first %d MiB of the same text, %d patterns of "
                                             "length %d, 1 thread, %.1f s" % (wsample >> 20, WM_PATTERNS, WM_LENGTH, wsecs))
        R.parity_ok = R.parity_ok and out["cpu_baseline_wm"]["counts_match"]
    # the legacy host-pointer path (search_ac as main.c calls it): the text is a pageable host buffer; PCIe-bound; never `value`
    hp = []
    for _ in range(4):
        t0 = time.perf_counter()
        legacy_cnt, ksecs = acs[AC_LENGTHS[0]].count_host(host_text, S.VARIANT_TUNED)
        hp.append(time.perf_counter() - t0)
    out["host_pointer_path"] = dict(what="smh_ac_count_host (what search_ac runs) on the whole %d MiB text in pageable host memory: 64 MiB "
                                         "pieces through two device buffers of a pooled workspace, copy of piece k+1 beside the scan of "
                                         "piece k, count read back (round 3: hipMalloc + one copy + kernel + hipFree, 36.6 GB/s)" % (per_gpu >> 20),
                                    GBps=round(per_gpu / min(hp[1:]) / 1e9, 2), seconds=round(min(hp[1:]), 4),
                                    first_call_GBps=round(per_gpu / hp[0] / 1e9, 2), kernel_seconds=round(ksecs, 5),
                                    count_matches=legacy_cnt == R.local_counts[0])
    R.parity_ok = R.parity_ok and out["host_pointer_path"]["count_matches"]
    S.lib.smh_host_path_release()
    if cpu.kind == "reference":
        secs, ok, wall = cpu.ac_all_cores_reference(pats, AC_PATTERNS, SIGMA, prefix, cpu_counts)
        allc = dict(value=round(8.0 * sample * len(pats) / secs / 1e9, 3), unit="Gbit/s", cores=cpu.all_cores, kind="reference",
                    cpu=cpu.model, host_cpus=cpu.host_cpus, cpu_quota=cpu_quota(), counts_match=ok,
                    sample="same sample as byte-range shards (main.c:467-477) on %d threads = every thread this process may run on "
                           "(%d CPUs in the machine, cgroup quota %s CPUs); time = slowest shard's search_ac per set, summed (%.2f s; %.1f s wall with "
                           "preproc_ac repeated per shard as every MPI rank of the reference does)"
                           % (cpu.all_cores, cpu.host_cpus, cpu_quota() if cpu_quota() is not None else "none", secs, wall))
        if R.wpat is not None:
            wsecs, wok = cpu.wm_all_cores_reference(R.wpat, WM_LENGTH, WM_PATTERNS, SIGMA, host_text[:wsample], wcnt)
            allc["wm"] = dict(value=round(8.0 * wsample / wsecs / 1e9, 3), unit="Gbit/s", counts_match=wok,
                              sample="search_wu2, the WM sample as %d byte-range shards, slowest shard %.2f s" % (cpu.all_cores, wsecs))
            ok = ok and wok
        out["cpu_baseline_all_cores"] = allc
        R.parity_ok = R.parity_ok and ok


def finish(R):
    """rank 0: the record out (compact line LAST on stdout, full record to bench_detail.json); every rank: the process group down"""
    if R.rank == 0:
        R.out["parity_ok"] = bool(R.parity_ok)
        R.out["smh_multi_ok"] = R.multi_ok
        R.out["wall_s"] = round(time.perf_counter() - R.wall_t0, 1)
        R.out["phases_s"] = R.phases
        emit(R.out)
    if R.world > 1:
        R.dist.barrier()
        R.dist.destroy_process_group()


def fail_run(R, why):
    """A parity failure ends the run -- through the same short last line as a good run (`parity_ok`: false, `error`), so that a
    driver that keeps an 8 KB tail of stdout can still parse it (round 5 printed the full 30 KB record here)."""
    R.parity_ok = False
    if R.rank == 0 and R.out is not None:
        R.out["error"] = "PARITY FAILURE: " + why
        R.out["parity_ok"] = False
        R.out["smh_multi_ok"] = R.multi_ok
        R.out["wall_s"] = round(time.perf_counter() - R.wall_t0, 1)
        emit(R.out)
    raise SystemExit("PARITY FAILURE: " + why)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib-per-gpu", type=int, default=1024, help="text bytes per GPU in MiB (BASELINE: 1024)")
    ap.add_argument("--cpu-sample-mib", type=int, default=96, help="prefix the serial search_ac baseline runs on")
    ap.add_argument("--cpu-wm-sample-mib", type=int, default=16, help="prefix the serial search_wu2 baseline runs on")
    ap.add_argument("--shard-mib", type=int, default=4096, help="per-GPU shard of the 32 GB configurations (configs[3], [4])")
    ap.add_argument("--verify-mib", type=int, default=-1,
                    help="MiB of every 4 GiB shard the CPU recounts (head + last 64 MiB); 0 = all of it; default: all at N = 1; at N > 1 8 MiB per host thread of the rank, 128..512")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines and the verification")
    ap.add_argument("--no-wm", action="store_true", help="skip the side configurations (WM, configs[3], configs[4])")
    ap.add_argument("--no-multi", action="store_true", help="skip the one-process smh_multi leg")
    ap.add_argument("--no-skewed", action="store_true", help="skip the non-uniform corpora (the `skewed` object)")
    ap.add_argument("--no-small", action="store_true", help="skip the reference-size texts and the preproc timings (`small_text`, `preproc`)")
    ap.add_argument("--multi-leg", type=int, default=0, help=argparse.SUPPRESS)  # internal: the child of the smh_multi leg
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal of the N > 1 control flow on ONE card: every rank uses device 0, process group over gloo")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0:
        raise SystemExit("bench.py: --steps must be >= 1 and --warmup >= 0")
    if args.multi_leg:
        return multi_leg(args)
    if os.environ.get("WORLD_SIZE") is None and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return

    R = Run(args)
    phase_headline(R)
    phase_stream_read_and_positions(R)
    if not args.no_wm:
        phase_wm_and_automaton(R)
        phase_mixed_lengths(R)
    R.mark('side measurements (stream read, positions, WM, automaton, mixed)')
    if not args.no_wm:
        phase_shard_configs(R)
    R.mark('32 GB configurations (configs[3], configs[4], more lengths)')
    if not args.no_wm and not args.no_skewed:
        phase_skewed(R)
    R.mark('skewed corpora')
    if not args.no_wm:
        phase_table_kernels(R)
    R.mark('table kernels')
    if not args.no_small:
        phase_small_text(R)
        phase_preproc(R)
    R.mark('reference-size texts, preproc')
    if not args.no_multi and not args.share_device:
        phase_multi_leg(R)
    R.mark('smh_multi leg')
    if not args.no_cpu:
        phase_verify(R)
    R.mark('verification')
    if R.rank == 0 and R.world == 1 and not args.no_cpu:
        phase_cpu_baselines(R)
    R.mark("cpu baselines, host-pointer path")
    finish(R)
    if not R.parity_ok:
        raise SystemExit("PARITY FAILURE: GPU counts differ from the CPU reference")
    if R.multi_ok is False and R.world == 1:
        # N = 1: a leg that does not run is a defect of this tree.  N > 1: the per-rank measurement above is complete and verified by
        # itself; the one-process side leg (a child that opens its own communicator over all devices beside the ranks') stays a flag
        # on the record (`smh_multi_ok: false`, `smh_multi.error`) and a line on stderr, not an exit code that discards the record
        raise SystemExit("bench.py: the smh_multi leg did not complete (smh_multi_ok false on the record); --no-multi skips it")


if __name__ == "__main__":
    main()
