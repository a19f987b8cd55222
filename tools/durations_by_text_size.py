#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 kernel trace, with the launches of one kernel instance over
different text sizes (1 GiB headline sets, 4 GiB shards) kept apart:

    python tools/durations_by_text_size.py gpurun_out/rNN/trace > profiles/rNN/kernel_durations_by_text_size.txt
"""
import collections
import csv
import glob
import sys

KEEP = ("ac_dfa", "wm_block", "wm_pair", "wm_gram", "acm_kernel", "ac_table", "wm_table", "stream_read")


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
    """bench.py's step = its three headline scans back to back; the first long run of launches in the trace is conditioning +
    warm-up + timed steps.  -> mean duration per kernel instance over the LAST `steps` steps of that run (the timed ones)."""
    rows = []
    for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if any(k in name for k in KEEP):
                rows.append((int(r["Start_Timestamp"]), name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    # the run: from the first launch, as long as the kernel names repeat with period 3
    run = []
    for i, (_, name, us) in enumerate(rows):
        if i >= 3 and name != rows[i - 3][1]:
            break
        run.append((name, us))
    if len(run) < 3 * steps:
        return
    last = run[len(run) - len(run) % 3 - 3 * steps: len(run) - len(run) % 3]
    print("\nthe step's launches: first run of %d launches (conditioning + warm-up + timed steps); its last %d steps:" % (len(run), steps))
    per = collections.OrderedDict()
    for name, us in last:
        per.setdefault(name, []).append(us)
    for name, v in per.items():
        print("%-62s n=%d avg %.1f us min %.1f max %.1f" % (name[:62], len(v), sum(v) / len(v), min(v), max(v)))
    print("whole run, per kernel: " + "  ".join("%s avg %.1f" % (n.split("<")[0] + "<" + n.split("<")[1][:14], sum(u for m, u in run if m == n) / sum(1 for m, u in run if m == n)) for n in per))


if __name__ == "__main__":
    main()
    timed_steps(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
