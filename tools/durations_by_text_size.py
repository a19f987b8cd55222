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


if __name__ == "__main__":
    main()
