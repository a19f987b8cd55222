"""bench.py's bookkeeping that needs no GPU: which kernel instance serves a handle at the Aho-Corasick entry point, and the
lookup of its measured HBM traffic in profiles/hbm_traffic.json (quoted only for the same build, per text size)."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _info(**kw):
    d = dict(m=16, scan_dense=0, scan_engine=0, verify_in_registers=0, gram_kind=1, scan_depth=12, scan_stride=2, lds_rows=7816,
             scan_full_rows=4817, scan_exact=0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_kernel_instance_names():
    assert bench.ac_kernel_name(_info(m=8, scan_dense=1)) == "wm_pair_kernel<false, 1024>"
    assert bench.ac_kernel_name(_info(scan_engine=1, verify_in_registers=1)) == "wm_gram_kernel<1, false, 5, false>"
    assert bench.ac_kernel_name(_info(m=32, scan_engine=1, verify_in_registers=1)) == "wm_gram_kernel<1, false, 6, false>"
    assert bench.ac_kernel_name(_info(m=32, scan_engine=1)) == "wm_gram_kernel<1, false, 2, false>"
    assert bench.ac_kernel_name(_info(scan_engine=1, verify_in_registers=1, gram_kind=5)) == "wm_gram_kernel<5, false, 5, false>"
    assert bench.ac_kernel_name(_info(scan_engine=1, verify_in_registers=2, gram_kind=5)) == "wm_gram_kernel<5, false, 3, false>"  # windows from L2 (round 6)
    assert bench.ac_kernel_name(_info(m=32, scan_engine=1, verify_in_registers=2, gram_kind=1)) == "wm_gram_kernel<1, false, 4, false>"
    assert bench.ac_kernel_name(_info()) == "ac_dfa_kernel<unsigned short, 4, 4, 1, false,"          # hybrid, depth-cut
    assert bench.ac_kernel_name(_info(scan_full_rows=0, scan_exact=1, scan_depth=8)) == "ac_dfa_kernel<unsigned short, 4, 2, 1, true,"


def test_traffic_is_quoted_per_text_size_and_only_for_the_same_build(monkeypatch):
    rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    monkeypatch.setattr(bench, "kernel_build_id", lambda: rec["build_id"])
    gib = 1 << 30
    for info in (_info(m=8, scan_dense=1), _info(scan_engine=1, verify_in_registers=1), _info(m=32, scan_engine=1, verify_in_registers=1)):
        name = bench.ac_kernel_name(info)
        if not any(k.startswith(name) for k in rec["kernels"]):
            pytest.skip("profiles/hbm_traffic.json predates kernel instance " + name)
        got, src = bench.measured_traffic(info, gib)
        assert got is not None and 1.0 * gib <= got < 1.1 * gib, (name, got, src)   # the 1 GiB launches, not the 4 GiB shards'
        # ... and the 4 GiB shards' figure where the instance also served one (else nothing is quoted for that size)
        entry = next(v for k, v in rec["kernels"].items() if k.startswith(name))
        has4 = any(3.9 * gib < g["hbm_read_bytes"] < 4.5 * gib for g in entry.get("by_text_size", []))
        got4, _ = bench.measured_traffic(info, 4 * gib)
        assert (got4 is not None and 4.0 * gib <= got4 < 4.4 * gib) if has4 else got4 is None, (name, got4)
    monkeypatch.setattr(bench, "kernel_build_id", lambda: "another build")
    got, src = bench.measured_traffic(_info(m=8, scan_dense=1), gib)
    assert got is None and "not quoted" in src


def test_last_line_is_short_and_carries_roofline_and_cpu_baseline():
    """The driver parses the LAST stdout line out of an 8 KB tail (round 4's 37.7 KB line came back as parsed = null): the
    compact line built from a full round-4 record must stay under 4 KB, round-trip through json and hold both objects."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_final", "bench.json")))
    line = bench.compact_line(rec, "bench_detail.json", "0" * 64)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 4096
    got = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in got, key
    assert got["value"] == rec["value"] and got["ms_per_step"] == rec["ms_per_step"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(got["roofline"])
    assert got["roofline"]["frac"] == rec["roofline"]["frac"] and got["roofline"]["automaton"]["frac"] == rec["roofline"]["automaton"]["frac"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(got["cpu_baseline"])
    assert got["config"]["workload"] and "model" not in got["config"]
    assert got["verified"]["all_equal"] is True and got["hbm_frac"]["wm_ascii"]["m5"] == rec["wm_ascii"]["m5"]["hbm_frac"]
    # a record bloated far beyond anything bench.py writes still yields a parseable short line with the mandatory objects
    fat = dict(rec)
    fat["skewed"] = dict(rec["skewed"], **{"corpus%d" % i: rec["skewed"]["dna_repeats"] for i in range(200)})
    fat["config"] = dict(rec["config"], workload="x" * 5000)
    line = bench.compact_line(fat, "bench_detail.json", "0" * 64)
    assert len(line) < 4096 and {"roofline", "cpu_baseline", "value"} <= set(json.loads(line))


def test_line_says_who_ran_and_whether_the_run_held():
    """Round 6: the N > 1 record carries the backend and world size torch.distributed reports
    and one card identity per rank; `parity_ok` / `smh_multi_ok` / `error` are in the part of the line that is never dropped."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_final", "bench_detail.json")))
    ranks = [dict(rank=i, local_rank=i, device=i, pci_bus_id="0000:%02x:00.0" % (0x10 + i), uuid="GPU-%d" % i, host="node", pid=100 + i) for i in range(8)]
    rec.update(parity_ok=True, smh_multi_ok=False, error="x" * 2000,
               world=dict(world_size=8, backend="nccl", distinct_cards=8, rehearsal=False, ranks=ranks))
    rec["skewed"] = dict(rec["skewed"], **{"corpus%d" % i: rec["skewed"]["dna_repeats"] for i in range(200)})  # bloat: optional groups go
    got = json.loads(bench.compact_line(rec, "bench_detail.json", "0" * 64))
    assert got["parity_ok"] is True and got["smh_multi_ok"] is False and len(got["error"]) == 300
    w = got["world"]
    assert w["world_size"] == 8 and w["backend"] == "nccl" and w["distinct_cards"] == 8 and w["rehearsal"] is False
    assert [r["pci_bus_id"] for r in w["ranks"]] == [r["pci_bus_id"] for r in ranks] and set(w["ranks"][0]) == {"rank", "local_rank", "device", "pci_bus_id"}
    # the six-rank one-card rehearsal: six identical bus ids, gloo, flagged
    reh = dict(rec, world=dict(world_size=6, backend="gloo", distinct_cards=1, rehearsal=True, ranks=[dict(r, local_rank=0, device=0, pci_bus_id="0000:75:00.0") for r in ranks[:6]]))
    w = json.loads(bench.compact_line(reh))["world"]
    assert w["backend"] == "gloo" and w["rehearsal"] is True and w["distinct_cards"] == 1 and len({r["pci_bus_id"] for r in w["ranks"]}) == 1


def test_sharding_label_names_the_collective_that_ran():
    assert "no collective" in bench.sharding_label(1, None, False)
    assert "RCCL" in bench.sharding_label(8, "nccl", False) and "x8" in bench.sharding_label(8, "nccl", False)
    lab = bench.sharding_label(6, "gloo", True)
    assert "gloo" in lab and "RCCL" not in lab and "REHEARSAL" in lab


def test_small_text_and_preproc_reach_the_line():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_final", "bench_detail.json")))
    rec["small_text"] = {"workload": "w", "E.coli": {"bytes": 4628736, "alphabet": 4, "ac_1000_m8": {"of_gib_rate": 0.2}, "wm_8000_m8": {"of_gib_rate": 0.19}}}
    rec["preproc"] = {"what": "w", "sets": {"ac_1000_m8": {"preproc_s": 0.01}}}
    got = json.loads(bench.compact_line(rec))
    assert got["small_text_of_gib_rate"] == {"E.coli": [0.2, 0.19]} and got["preproc_s"] == {"ac_1000_m8": 0.01}


def test_cards_that_share_a_bus_id_are_told_apart_by_uuid_and_the_line_says_so():
    """rank_identity counts (host, PCI bus id, UUID): logical devices of a partitioned card may report one bus id.  The compact
    line then carries the UUIDs as well, so that the reader can see why distinct_cards exceeds the distinct bus ids."""
    ranks = [dict(rank=i, local_rank=i, device=i, pci_bus_id="0000:75:00.0", uuid="GPU-%d" % i, host="node", pid=100 + i) for i in range(2)]
    rec = dict(metric="Gbit/s", value=1.0, unit="Gbit/s", n_gpus=2, steps=1, warmup=0, ms_per_step=1.0, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="u8", data="synthetic", config=dict(workload="w"), parity_ok=True, smh_multi_ok=True,
               roofline=dict(bound="hbm", achieved=1.0, peak=8000.0, unit="GB/s", frac=0.1, traffic=None),
               cpu_baseline=dict(value=1.0, unit="Gbit/s", cores=1, kind="reference", sample="s"),
               world=dict(world_size=2, backend="nccl", distinct_cards=2, rehearsal=False, ranks=ranks))
    w = json.loads(bench.compact_line(rec))["world"]
    assert w["distinct_cards"] == 2 and [r["uuid"] for r in w["ranks"]] == ["GPU-0", "GPU-1"]
