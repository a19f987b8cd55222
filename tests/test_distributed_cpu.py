"""N > 1 path on the CPU: world_size-2 (and 3) process groups over gloo run the same shard + reduce
code bench.py uses under RCCL (cuda-aho-corasick-wu-manber_amd/sharded.py).  Each rank scans
only its own byte range -- with the CPU lane emulator standing in for the GPU -- and one
all-reduce of the 64-bit count must reproduce the reference's whole-text count."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
    VECTORS = {v["name"]: v for v in json.load(f)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_lib as E
    from emu_lib import S
    sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
    import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vec = VECTORS[name]
    text, pat = cases.build(vec)
    m, p, sigma = vec["m"], vec["p"], vec["sigma"]
    b, e = sharded.shard_for_rank(len(text), world, rank, m)
    ac = S.AcAutomaton.from_patterns(pat, m, p, sigma)
    wm = S.WmTables.from_patterns(pat, m, p, sigma)
    local = torch.tensor([E.ac_scan(ac, text[b:e], 0, 1), E.wm_scan(wm, text[b:e], 0, 1)], dtype=torch.int64)
    mine = local.clone()
    everyone = sharded.gather_counts(local)  # [world, 2]: what bench.py reports as per-GPU counts
    assert everyone.shape == (world, 2) and torch.equal(everyone[rank], mine)
    # the overlapped form bench.py times: every step reduces its own buffer, all are waited for at the end
    steps = [local.clone() for _ in range(3)]
    sharded.finish([sharded.reduce_count_async(c) for c in steps])
    sharded.reduce_count(local)
    assert torch.equal(everyone.sum(dim=0), local) and all(torch.equal(c, local) for c in steps)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([mine[0], mine[1], local[0], local[1], b, e]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "kat_1m003_r3"), (2, "edge_n8207"), (3, "dense_dna"), (2, "n_eq_m")])
def test_sharded_counts_sum_to_the_reference_count(world, name, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    want = VECTORS[name]["count_ac"]
    per_rank = [np.load(os.path.join(str(tmp_path), "r%d.npy" % r)) for r in range(world)]
    for r in per_rank:
        assert int(r[2]) == want and int(r[3]) == want      # every rank holds the reduced total
    assert sum(int(r[0]) for r in per_rank) == want           # AC shard counts
    assert sum(int(r[1]) for r in per_rank) == want           # WM shard counts
    # shards tile the text with an m-1 overlap (main.c:467-477)
    m = VECTORS[name]["m"]
    for a, b in zip(per_rank[:-1], per_rank[1:]):
        assert int(a[5]) - int(b[4]) == min(m - 1, int(a[5]) - int(b[4])) and int(b[4]) <= int(a[5])


# ---------------------------------------------------------------------------------------------------------
# The 32 GB configurations' N > 1 flow (bench.py `ac_8000_patterns` / `wm_ascii` at --gpus N) at small size: the same
# helpers (sharded.shard_plan / reduce_count / summarize_shard_runs / verify_slices / merge_verified / host_barrier),
# the CPU lane emulator standing in for the GPU scan, gloo for RCCL.
CONFIGS = [("ac_many", "ac", 4, (8, 16, 32), 300, 10), ("wm_bytes", "wm", 256, (5, 8, 12, 20), 800, 9)]
SHARD = 196_613  # bytes per rank; not a multiple of anything the kernels tile by


def _worker_configs(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_lib as E
    from emu_lib import S
    import oracle_lib as O
    sys.path.insert(0, os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd"))
    import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_tot = SHARD * world
    report, mine = {}, {}
    for label, algo, sigma, lengths, p, seed in CONFIGS:
        begin, resident, lens = sharded.shard_plan(n_tot, world, rank, lengths)
        assert begin == rank * SHARD and resident == min(SHARD + max(lengths) - 1, n_tot - begin)
        text = S.corpus_text(resident, 42, sigma, offset=begin)  # this rank's byte range only
        for m in lengths:
            pat = S.corpus_patterns(m, p, seed, sigma, 42, n_tot, 2)
            h = (S.AcAutomaton if algo == "ac" else S.WmTables).from_patterns(pat, m, p, sigma)
            scan = (lambda t, h=h: E.ac_scan(h, t, 0, 1)) if algo == "ac" else (lambda t, h=h: E.wm_scan(h, t, 0, 1))
            local = scan(text[:lens[m]])
            cnt = torch.tensor([local], dtype=torch.int64)
            sharded.reduce_count(cnt)
            recs = sharded.gather_objects(dict(n=lens[m], ms=1.0 + rank, matches=local))
            obj = sharded.summarize_shard_runs(recs, 8000.0)
            assert obj["matches"] == int(cnt.item()) and len(obj["per_gpu_matches"]) == world
            assert obj["kernel_ms"] == float(world) and obj["per_gpu_ms"] == [1.0 + r for r in range(world)]
            report["%s.m%d" % (label, m)] = obj
            # what bench.py's `verified` does per rank: a head and a tail slice of the shard, GPU (here: emulator) against the oracle
            slices = sharded.verify_slices(lens[m], m, 96 << 10, tail=32 << 10)
            assert len(slices) == 2 and slices[1][0] % 16 == 0 and slices[1][0] + slices[1][1] == lens[m]
            g = [scan(text[o:o + l]) for o, l in slices]
            c = [(O.oracle_ac if algo == "ac" else O.oracle_wu)(pat, m, p, sigma, text[o:o + l])[0] for o, l in slices]
            mine["%s.m%d" % (label, m)] = dict(gpu=g, cpu=c, slices=[list(s) for s in slices], shard_bytes=lens[m])
    if rank == 1:
        mine["only_on_rank_1"] = dict(gpu=[1], cpu=[1], slices=[[0, 1]], shard_bytes=1)  # a name a rank lacks is a failure
    merged, ok = sharded.merge_verified(sharded.gather_objects(mine))
    sharded.host_barrier("test_configs")  # the store barrier bench.py parks the ranks in during the one-process leg
    if rank == 0:
        with open(os.path.join(out_dir, "report.json"), "w") as f:
            json.dump(dict(report=report, merged=merged, ok=ok), f)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_configurations_flow_and_per_shard_verification(world, tmp_path):
    import oracle_lib as O
    from emu_lib import S
    mp.spawn(_worker_configs, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rep = json.load(open(os.path.join(str(tmp_path), "report.json")))
    n_tot = SHARD * world
    for label, algo, sigma, lengths, p, seed in CONFIGS:
        whole = S.corpus_text(n_tot, 42, sigma)
        for m in lengths:
            pat = S.corpus_patterns(m, p, seed, sigma, 42, n_tot, 2)
            want = (O.oracle_ac if algo == "ac" else O.oracle_wu)(pat, m, p, sigma, whole)[0]
            obj = rep["report"]["%s.m%d" % (label, m)]
            assert obj["matches"] == want and sum(obj["per_gpu_matches"]) == want and want > 0
            v = rep["merged"]["%s.m%d" % (label, m)]
            assert v["equal"] and len(v["per_gpu"]) == world and all(e["gpu"] == e["cpu"] for e in v["per_gpu"])
    assert not rep["ok"] and not rep["merged"]["only_on_rank_1"]["equal"]  # the deliberately lopsided entry is caught
    assert all(v["equal"] for k, v in rep["merged"].items() if k != "only_on_rank_1")
