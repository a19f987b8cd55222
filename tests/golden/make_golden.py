"""Generates tests/golden/ref_vectors.json by RUNNING THE REFERENCE ITSELF.

Run in the authoring container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

For every seeded case of tests/cases.py it calls the reference's own compiled
ac/ac.c, wu/wu.c, sh/sh.c and sbom/sbom.c (oracle/_ref/libref.so; build recipe oracle/Makefile) and
records what they produce: the match counts of search_ac / search_wu /
search_wu2 / search_sh / search_sbom, struct ac_table.idcounter / patterncounter, and FNV-1a digests of
every table the reference fills (state_transition, state_supply, state_final,
SHIFT, PREFIX_size, PREFIX_value / PREFIX_index bucket heads).  Inputs are not
stored: they are regenerated from the seeds.  The file is data (expected
outputs), not reference source.
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
    for case in cases.all_cases():
        text, pat = cases.build(case)
        n, p, m, sigma = case["n"], case["p"], case["m"], case["sigma"]
        c_ac, t_ac, _, _ = O.ref_ac(pat, m, p, sigma, text)
        c_w2, t_w2, _, _ = O.ref_wu(pat, m, p, sigma, text, flat=True)
        c_w1, t_w1, _, _ = O.ref_wu(pat, m, p, sigma, text, flat=False)
        assert t_w1.digest() == t_w2.digest()
        # Set-Horspool (sh/sh.c); bmBc is the oracle's ora_pre_bmbc -- upstream's preBmBc is in its missing helper
        c_sh, t_sh = O.ref_sh(pat, m, p, sigma, text)
        # Set Backward Oracle Matching (sbom/sbom.c)
        c_sb, t_sb = O.ref_sbom(pat, m, p, sigma, text)
        rec = dict(case)
        rec.update(count_ac=c_ac, count_wu=c_w1, count_wu2=c_w2, idcounter=t_ac.idcounter,
                   patterncounter=t_ac.patterncounter,
                   fnv_transition="%016x" % O.fnv(t_ac.state_transition[:t_ac.idcounter * sigma]),
                   fnv_supply="%016x" % O.fnv(t_ac.state_supply[:t_ac.idcounter]),
                   fnv_final="%016x" % O.fnv(t_ac.state_final[:t_ac.idcounter]),
                   fnv_wm=["%016x" % d for d in t_w2.digest()],
                   count_sh=c_sh, sh_idcounter=t_sh.idcounter, sh_patterncounter=t_sh.patterncounter,
                   fnv_sh_transition="%016x" % O.fnv(t_sh.state_transition[:t_sh.idcounter * sigma]),
                   fnv_sh_final="%016x" % O.fnv(t_sh.state_final[:t_sh.idcounter]),
                   fnv_bmbc="%016x" % O.fnv(t_sh.bmBc),
                   count_sbom=c_sb, sbom_idcounter=t_sb.idcounter, sbom_patterncounter=t_sb.patterncounter,
                   fnv_sbom_transition="%016x" % O.fnv(t_sb.state_transition[:t_sb.idcounter * sigma]),
                   fnv_sbom_final="%016x" % O.fnv(t_sb.state_final_multi[:t_sb.idcounter * 200]))
        out.append(rec)
        print(case["name"], c_ac, c_w1, c_w2, t_ac.idcounter, flush=True)
    with open(os.path.join(HERE, "ref_vectors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "vectors")


if __name__ == "__main__":
    main()
