#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d FETCH_DIR -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d WRITE_DIR -- python3 bench.py --steps 3 --warmup 1 --no-cpu
    python tools/make_traffic_json.py FETCH_DIR WRITE_DIR > profiles/rNN/hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so the
read figure is doubled (the guide's correction).  Values are means over the dispatches of each kernel."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_build_id():
    """Same digest as bench.py: the sources the library under test was built from."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".inc", ".hip", ".c")):
            with open(os.path.join(src, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:12]


def load(path, counter):
    agg = collections.OrderedDict()
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if not any(k in name for k in ("ac_dfa", "wm_block", "wm_pair", "wm_gram", "acm_kernel", "ac_table", "wm_table", "stream_read", "key_kernel", "hash_kernel")):
                continue
            agg.setdefault(name, []).append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = collections.OrderedDict()
    for name, vals in fetch.items():
        fk = sum(vals) / len(vals)
        wv = write.get(name, [0.0])
        wk = sum(wv) / len(wv)
        rd, wr = int(round(fk * 1024 * 2)), int(round(wk * 1024))
        out[name] = dict(FETCH_SIZE_KB_mean=fk, dispatches=len(vals), WRITE_SIZE_KB_mean=wk,
                         hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr)
        # one kernel instance may serve launches over different text sizes (1 GiB headline sets, 4 GiB shards):
        # dispatches whose read volume differs by more than 1.5x are reported as separate groups
        groups, cur = [], []
        for v in sorted(vals):
            if cur and v > 1.5 * cur[0]:
                groups.append(cur)
                cur = []
            cur.append(v)
        groups.append(cur)
        if len(groups) > 1:
            out[name]["by_text_size"] = [dict(dispatches=len(g), hbm_read_bytes=int(round(sum(g) / len(g) * 1024 * 2))) for g in groups]
    json.dump(dict(build_id=kernel_build_id(), profile=sys.argv[3] if len(sys.argv) > 3 else "",
                   source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
                          "--warmup 1 --no-cpu; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests "
                          "as 64 B); AC and wm_pair kernels of the headline sets scan 1 GiB per launch, the configs[3] / configs[4] "
                          "instances the shard size bench.py reports",
                   kernels=out), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
