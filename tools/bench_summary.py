#!/usr/bin/env python3
"""tools/bench_summary.py FILE -- the last JSON line of a bench.py run, one object per line (for reading a gpurun tail)."""
import json
import sys

lines = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
if not lines:
    sys.exit("no JSON line in " + sys.argv[1])
d = json.loads(lines[-1])
print("value", d["value"], d["unit"], "n_gpus", d["n_gpus"], "ms_per_step", d["ms_per_step"], "build", d.get("kernel_build_id"))
print("roofline", d["roofline"])
print("after_idle", d.get("after_idle"))
for k in ("ac", "ac_automaton", "wm", "wm_long", "mixed_8_32", "ac_8000_patterns", "wm_ascii", "wm_ascii_more", "table_kernels", "smh_multi"):
    v = d.get(k)
    if not isinstance(v, dict):
        continue
    print(k + ":", v.get("error", ""))
    for kk, vv in v.items():
        if isinstance(vv, dict):
            print("   ", kk, {a: b for a, b in vv.items() if a not in ("per_gpu_matches", "per_gpu_ms")})
for k in ("stream_read", "positions", "cpu_baseline", "cpu_baseline_wm", "cpu_baseline_all_cores", "host_pointer_path", "parity"):
    if k in d:
        print(k, d[k])
sk = d.get("skewed")
if isinstance(sk, dict):
    print("skewed: worst chosen_vs_best_forced", sk.get("worst_chosen_vs_best_forced"))
    for corpus, sets in sk.items():
        if not isinstance(sets, dict):
            continue
        for name, r in sets.items():
            print("   ", corpus, name, "chosen:", r["chosen"]["engine"], r["chosen"]["kernel_ms"], "ms", r["chosen"]["hbm_frac"], "first launch", r["chosen"].get("first_launch_ms"), "ms; engines per launch", r["chosen"]["engines_per_launch"],
                  "| forced:", {k: v["kernel_ms"] for k, v in r["forced"].items()}, "| ratio", r.get("chosen_vs_best_forced"), "agree", r["engines_agree"])
v = d.get("verified")
if v:
    print("verified all_equal", v["all_equal"], "seconds", v["seconds"], "names", len(v["counts"]))
    for name, e in v["counts"].items():
        if not e["equal"]:
            print("  MISMATCH", name, e)
