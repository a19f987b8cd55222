"""Generates tests/golden/ref_sog_vectors.json by RUNNING THE REFERENCE ITSELF (sog/sog8.c compiled into
oracle/_ref/libref.so by oracle/Makefile).

    python tests/golden/make_golden_sog.py      (needs /root/reference and `make -C oracle ref`)

Recorded per case: FNV digests of the tables preproc_sog8 fills deterministically -- T8 (2^24 bytes), the sorted
pattern hashes scanner_hs and the permutation scanner_index -- and `count`: the reference's search_ac on the same
text and the same 8-byte patterns, i.e. the number of 8-byte windows that equal a pattern, which is the quantity
search_sog8 is meant to return.  The reference's OWN search_sog8 count is recorded too (`count_ref_sog8`, not
asserted anywhere): its 2-level bitmap is computed from an uninitialised variable (sog/sog8.c:124,135), so it drops
matches depending on stack contents.  Inputs are regenerated from the seeds in tests/cases.py.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = []
    for case in cases.sog_cases():
        text, pat = cases.build(case)
        p, sigma = case["p"], case["sigma"]
        tabs = O.SogTables(p)
        ref_cnt = O.ref_sog8(pat, p, text, tabs)
        count, _, _, _ = O.ref_ac(pat, 8, p, sigma, text)
        rec = dict(case)
        rec.update(fnv_T8="%016x" % O.fnv(tabs.T8), fnv_hs="%016x" % O.fnv(tabs.scanner_hs),
                   fnv_index="%016x" % O.fnv(tabs.scanner_index), count=count, count_ref_sog8=ref_cnt)
        out.append(rec)
        print(case["name"], count, ref_cnt, flush=True)
    with open(os.path.join(HERE, "ref_sog_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
