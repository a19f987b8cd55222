#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 kernel trace, with the launches of one kernel instance over
different text sizes (1 GiB headline sets, 4 GiB shards) kept apart:

    python tools/durations_by_text_size.py gpurun_out/rNN/trace > profiles/rNN/kernel_durations_by_text_size.txt
"""
import collections
import csv
import glob
import sys

KEEP = ("ac_dfa", "wm_block", "wm_pair", "wm_gram", "acm_kernel", "key_kernel", "hash_kernel", "ac_table", "wm_table", "stream_read")


def main():
    d = collections.OrderedDict()
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if any(k in name for k in KEEP):
                d.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("per-dispatch durations from the kernel trace of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu`,")
    print("grouped where one kernel instance served launches over different text sizes (1 GiB headline sets, 4 GiB shards)\n")
    for name, v in d.items():
        groups, cur = [], []
        for x in sorted(v):
            if cur and x > 2.5 * cur[0]:
                groups.append(cur)
                cur = []
            cur.append(x)
        groups.append(cur)
        print("%-62s %s" % (name[:62], "  ".join("n=%d avg %.1f us min %.1f" % (len(g), sum(g) / len(g), g[0]) for g in groups)))


def timed_steps(path, steps=10):
    """bench.py's step = its three headline scans back to back.  The first run of launches in the trace is conditioning +
    warm-up + timed steps; after an idle gap the same warm-up + timed steps follow once more (`after_idle`).  -> mean duration
    per kernel instance over the LAST `steps` steps of each of the two runs."""
    rows = []
    for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if any(k in name for k in KEEP):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # the launches whose kernel names repeat with period 3, split where the device sat idle for more than 50 ms
    runs, cur = [], []
    for i, (t0, t1, name) in enumerate(rows):
        if i >= 3 and name != rows[i - 3][2]:
            break
        if cur and t0 - rows[i - 1][1] > 50_000_000:
            runs.append(cur)
            cur = []
        cur.append((name, (t1 - t0) / 1e3))
    runs.append(cur)
    labels = ["conditioning + warm-up + timed steps (the line's `value`)", "after an idle gap: warm-up + timed steps (`after_idle`)"]
    for run, label in zip(runs[:2], labels):
        if len(run) < 3 * steps:
            continue
        last = run[len(run) - len(run) % 3 - 3 * steps: len(run) - len(run) % 3]
        print("\n%s: %d launches; its last %d steps:" % (label, len(run), steps))
        per = collections.OrderedDict()
        for name, us in last:
            per.setdefault(name, []).append(us)
        for name, v in per.items():
            print("%-62s n=%d avg %.1f us min %.1f max %.1f" % (name[:62], len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
    timed_steps(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
