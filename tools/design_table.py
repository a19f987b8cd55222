#!/usr/bin/env python3
"""tools/design_table.py DIR -- the "Measured" table of DESIGN.md section 4 from the artefacts of one final job (bench_detail.json,
hbm_traffic.json, kernel_durations_by_text_size.txt under DIR), written between the markers `<!-- measured:begin -->` and
`<!-- measured:end -->` of DESIGN.md: the table cannot drift from the record it cites."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_final")
b = json.load(open(os.path.join(D, "bench_detail.json")))
h = json.load(open(os.path.join(D, "hbm_traffic.json")))
G = 1 << 30
dur = {}
for ln in open(os.path.join(D, "kernel_durations_by_text_size.txt")):
    mt = re.match(r"^(\S.*?)\s+n=", ln)
    if mt:
        dur[mt.group(1).strip()] = [(int(n), float(a)) for n, a in re.findall(r"n=(\d+) avg ([0-9.]+) us", ln)]


def trace_ms(prefix, big=False):
    """rocprofv3 mean of the instance's launches over the 1 GiB (or, big, the 4 GiB) texts"""
    for name, groups in dur.items():
        if name.startswith(prefix):
            full = [g for g in groups if g[0] >= 5]
            if full:
                g = full[-1] if big and len(full) > 1 else full[0] if not big else full[-1]
                return "%.4f" % (g[1] / 1e3)
    return "—"


def traffic(prefix, text_gib):
    for name, k in h["kernels"].items():
        if name.startswith(prefix):
            groups = k.get("by_text_size") or [dict(dispatches=k["dispatches"], hbm_read_bytes=k["hbm_read_bytes"])]
            near = [g for g in groups if 0.9 * text_gib * G < g["hbm_read_bytes"] < (2.2 if text_gib > 1 else 1.5) * text_gib * G]
            if near:
                return " / ".join("%.2f" % (g["hbm_read_bytes"] / (text_gib * G)) for g in sorted(near, key=lambda g: g["hbm_read_bytes"]))
    return "—"


def fr(x):
    return "%.3f" % x


ac, a8, wa, wm = b["ac"], b["ac_8000_patterns"], b["wm_ascii"], b["wm_ascii_more"]
sk = b["skewed"]
rows = []
rows.append(("streaming-read probe, 1 GiB / 4 GiB", "`smh_stream_read_kernel`", "%.4f / %.3f ms [%s]" % (b["stream_read"]["ms"], b["stream_read"]["shard"]["ms"], trace_ms("smh_stream_read_kernel")),
             "%.2f / %.2f" % (b["stream_read"]["hbm_frac"], b["stream_read"]["shard"]["hbm_frac"]), "1.00"))
for m, bold in ((8, True), (16, True), (32, True)):
    r = ac["m%d" % m]
    inst = r["kernel_instance"]
    rows.append(("**configs[1] AC 1000 × %d**%s" % (m, " (`roofline`)" if inst == b["roofline"]["kernel_instance"] else ""), "`%s`" % inst,
                 "%.4f ms [%s]" % (r["kernel_ms"], trace_ms(inst)), "**%s**" % fr(r["hbm_frac"]), traffic(inst, 1)))
rows.append(("step = the three + count all-reduce", "—", "%.4f ms" % b["ms_per_step"], "**%d Gbit/s** (`value`)" % round(b["value"]), "—"))
au = b["ac_automaton"]
rows.append(("the same m = 16 / 32 on the automaton", "`ac_dfa_kernel<u16,4,4,1,false,3,…>`", "%.4f / %.4f ms [%s]" % (au["m16"]["kernel_ms"], au["m32"]["kernel_ms"], trace_ms("ac_dfa_kernel<unsigned short, 4, 4, 1, false, 3")),
             "%.2f / %.2f" % (au["m16"]["hbm_frac"], au["m32"]["hbm_frac"]), traffic("ac_dfa_kernel<unsigned short, 4, 4, 1, false, 3", 1)))
rows.append(("positions, m = 16", "`wm_gram_kernel<1, true, 5, false>`", "%.4f ms" % b["positions"]["kernel_ms"], "%.2f" % (G / b["positions"]["kernel_ms"] / 8e9), traffic("wm_gram_kernel<1, true, 5", 1)))
rows.append(("configs[2] WM 10 000 × 8", "`wm_pair_kernel<false, 1024>`", "%.4f ms" % b["wm"]["kernel_ms"], "%.2f" % b["wm"]["hbm_frac"], traffic("wm_pair_kernel", 1)))
mx = b["mixed_8_32"]
rows.append(("configs[1] as ONE set, lengths 8..32 (AC / WM entry)", "`acm_kernel<unsigned int, 4>` [%s] / grouped `wm_gram_kernel<4, …>` [%s]" % (trace_ms("acm_kernel"), trace_ms("wm_gram_kernel<4")),
             "%.4f / %.4f ms" % (mx["ac"]["kernel_ms"], mx["wm"]["kernel_ms"]), "%.2f / %.2f" % (mx["ac"]["hbm_frac"], mx["wm"]["hbm_frac"]), "%s / %s" % (traffic("acm_kernel", 1), traffic("wm_gram_kernel<4", 1))))
rows.append(("configs[3] AC 8000 × 8 / 16 / 32, 4 GiB", " / ".join("`%s` [%s]" % (a8["m%d" % m]["kernel_instance"], trace_ms(a8["m%d" % m]["kernel_instance"], True)) for m in (8, 16, 32)),
             " / ".join("%.3f" % a8["m%d" % m]["kernel_ms"] for m in (8, 16, 32)) + " ms", " / ".join("%.2f" % a8["m%d" % m]["hbm_frac"] for m in (8, 16, 32)),
             " / ".join(traffic(a8["m%d" % m]["kernel_instance"], 4).split(" / ")[-1] for m in (8, 16, 32))))
allw = dict(wa, **wm)
for label, ms_, inst in (("configs[4] WM 100 000, m = 5, 4 GiB", (5,), "wm_gram_kernel<10, false, 3"), ("configs[4] m = 6 / 7 / 8 / 9, 4 GiB", (6, 7, 8, 9), "wm_gram_kernel<9, false, 3"),
                         ("configs[4] m = 10 / 12 / 16, 4 GiB", (10, 12, 16), "wm_gram_kernel<8, false, 3"), ("configs[4] m = 20, 4 GiB", (20,), "wm_gram_kernel<8, false, 4")):
    rows.append((label, "`%s, false>` [%s mean]" % (inst, trace_ms(inst, True)), " / ".join("%.3f" % allw["m%d" % m]["kernel_ms"] for m in ms_) + " ms",
                 " / ".join(fr(allw["m%d" % m]["hbm_frac"]) for m in ms_),
                 traffic(inst, 4).replace(" / ", " (cuckoo verify entries) / ") + (" (m = 6: bucket table)" if " / " in traffic(inst, 4) else "")))
for label, corpus, sets in (("skewed DNA 8000 × 16 / 32", "dna_repeats", ("ac_8000_m16", "ac_8000_m32")), ("skewed proteins 10 000 × 8 / 1000 × 8 (round 5: 0.449 / 0.452 ms)", "protein_skewed", ("wm_10000_m8", "ac_1000_m8")),
                            ("skewed bytes 100 000 × 8 / 12 / 20 (150–230 true matches per 4 KiB)", "ascii_skewed", ("wm_100000_m8", "wm_100000_m12", "wm_100000_m20")),
                            ("uniform proteins 10 000 × 8 (round 5: key table, 0.453 ms)", "protein_uniform", ("wm_10000_m8",)),
                            ("skewed DNA 1000 × 8 / 16", "dna_repeats", ("ac_1000_m8", "ac_1000_m16"))):
    ch = [sk[corpus][s]["chosen"] for s in sets]
    rows.append((label, " / ".join(sorted({c["engine"] for c in ch}, key=[c["engine"] for c in ch].index)), " / ".join("%.3f" % c["kernel_ms"] for c in ch) + " ms",
                 " / ".join(fr(c["hbm_frac"]) for c in ch), "—"))
tk = b["table_kernels"]
rows.append(("reference tables walked as given (64 MiB)", "`ac_ / wm_ / sh_ / sbom_ / sog_table_kernel`", " / ".join("%.2f" % v["kernel_ms"] for v in tk.values() if isinstance(v, dict)) + " ms", "0.009 … 0.001", "1.5–1.8"))
st = b["small_text"]
names = ("world192", "E.coli", "A.thaliana.fna", "swiss-prot")
rows.append(("reference data-set sizes: " + " / ".join("%.1f" % (st[n]["bytes"] / 1e6) for n in names) + " MB (AC 1000 × 8)", "the handle's own kernel",
             " / ".join("%.1f" % (1e3 * st[n]["ac_1000_m8"]["kernel_ms"]) for n in names) + " µs", " / ".join("%.2f" % st[n]["ac_1000_m8"]["hbm_frac"] for n in names), "—"))
hp = b["host_pointer_path"]
rows.append(("`search_ac` on a host pointer, 1 GiB", "copy ∥ scan", "%.1f ms" % (1e3 * hp["seconds"]), "%.0f GB/s (PCIe)" % hp["GBps"], "—"))
rows.append(("CPU: the reference's `search_ac`, 1 thread / all the pool allows (quota %s)" % b["cpu_baseline_all_cores"].get("cpu_quota"), "`oracle/_ref/libref.so`", "—",
             "%.2f / %.1f Gbit/s" % (b["cpu_baseline"]["value"], b["cpu_baseline_all_cores"]["value"]), "—"))
out = ["<!-- measured:begin (tools/design_table.py over %s, build %s) -->" % (os.path.relpath(D, ROOT), b["kernel_build_id"]),
       "| Workload (per GPU) | kernel [rocprofv3 mean, ms] | time per launch (bench's events) | of 8 TB/s | HBM traffic ÷ algorithmic |", "|---|---|---|---|---|"]
out += ["| " + " | ".join(r) + " |" for r in rows]
v = b["verified"]
out.append("")
out.append("The run recounted %d counts on the CPU (%.0f s of its %.0f s wall): %s.  Traffic: `hbm_traffic.json` of build `%s`."
           % (len(v["counts"]), v["seconds"], b["wall_s"], "all equal" if v["all_equal"] else "MISMATCH", h["build_id"]))
out.append("<!-- measured:end -->")
text = "\n".join(out)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
if "<!-- measured:begin" in s:
    i, j = s.index("<!-- measured:begin"), s.index("<!-- measured:end -->") + len("<!-- measured:end -->")
    open(path, "w").write(s[:i] + text + s[j:])
    print("DESIGN.md updated (%d rows, build %s)" % (len(rows), b["kernel_build_id"]))
else:
    print(text)
