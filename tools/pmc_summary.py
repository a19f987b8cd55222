#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import collections
import csv
import glob
import sys

for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            name = r["Kernel_Name"]
            if not any(k in name for k in ("ac_dfa", "wm_block", "wm_pair", "wm_gram", "acm_kernel", "key_kernel", "keyb_kernel", "hash_kernel", "k_", "stream_read")):
                continue
            short = name.split("(")[0].replace("void ", "")
            agg.setdefault(short, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, ctrs in agg.items():
            print(k)
            for c, v in ctrs.items():
                print("    %-26s n=%-3d mean %.4g" % (c, len(v), sum(v) / len(v)))
