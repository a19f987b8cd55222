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
