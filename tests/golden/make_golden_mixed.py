"""Generates tests/golden/ref_mixed_vectors.json by RUNNING THE REFERENCE ITSELF, once per length class.

    python tests/golden/make_golden_mixed.py      (needs /root/reference and `make -C oracle ref`)

The reference API takes a single pattern length (smatcher.h:89,101), so a mixed-length set is defined
by its length-class decomposition: for every distinct length the reference's compiled search_ac and
search_wu2 (oracle/_ref/libref.so) are run on that class alone; the per-class counts and their sum
are recorded.  Wu-Manber needs m >= 3 (wu/wu.c:119-125): shorter classes have count_wu2 = null.
Inputs are regenerated from the seeds in tests/cases.py; the file holds expected outputs only.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = []
    for case in cases.mixed_cases():
        text, patterns, lengths = cases.build_mixed(case)
        per = []
        for L, flat in sorted(cases.split_classes(patterns, lengths).items()):
            p = len(flat) // L
            c_ac, t_ac, _, _ = O.ref_ac(flat, L, p, case["sigma"], text)
            c_wu = O.ref_wu(flat, L, p, case["sigma"], text, flat=True)[0] if L >= 3 else None
            assert c_wu is None or c_wu == c_ac
            per.append(dict(length=L, patterns=p, count_ac=c_ac, count_wu2=c_wu, distinct=t_ac.patterncounter))
        rec = dict(case)
        rec.update(per_class=per, total=sum(c["count_ac"] for c in per))
        out.append(rec)
        print(case["name"], rec["total"], [(c["length"], c["count_ac"]) for c in per], flush=True)
    with open(os.path.join(HERE, "ref_mixed_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
