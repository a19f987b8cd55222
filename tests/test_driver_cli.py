"""The `smatcher` command (csrc/smatcher_main.c): the reference driver's role for the AC / WM path --
main.c's arguments, data files, table setup, rank shards and report lines.
CPU: data creation, FASTA / protein decoding and table building (`-dry`), and the loud failure of the
search without a GPU.  GPU (-m gpu): the same translation unit built as smatcher_main.so is called
in-process and its totals equal the oracle's counts on the decoded text, for 1 and for 3 ranks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd")
sys.path.insert(0, PKG)
import smatcher_hip as S  # noqa: E402

EXE = os.path.join(PKG, "smatcher")
DSO = os.path.join(PKG, "smatcher_main.so")
no_spawn = pytest.mark.skipif(os.path.exists("/dev/kfd"),
                              reason="spawns a process; only run where no GPU can have been initialised")


def fnv1a64(a):
    h = 0xcbf29ce484222325
    for b in a.tobytes():
        h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def write_fasta(path, symbols, letters, width=70, junk="NNNN"):
    """symbols -> FASTA with two records, mixed case, N runs and blank lines thrown in"""
    s = "".join(letters[v] for v in symbols)
    half = len(s) // 2
    with open(path, "w") as f:
        f.write(">seq1 ACGT in the header must not count\n")
        body = s[:half].lower() + junk + s[half:]
        for i in range(0, len(body), width):
            f.write(body[i:i + width] + "\n")
            if i == 10 * width:
                f.write("\n>seq2 second record\n")


def run(args, cwd):
    return subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=300)


@no_spawn
def test_help_and_argument_errors(tmp_path):
    r = run(["--help"], tmp_path)
    assert r.returncode == 0 and "Usage: smatcher <ac|sh|sbom|wm|all> -m <m> -p_size <p_size> -n <n> -alphabet <alphabet>" in r.stdout
    assert run(["ac", "-m", "8"], tmp_path).stdout.startswith("smatcher - ")  # main.c:364-365: missing arguments -> usage
    r = run(["wm", "-m", "8", "-p_size", "100001", "-n", "1000", "-alphabet", "4"], tmp_path)
    assert r.returncode == 1 and "Only up to 100.000 patterns are supported" in r.stderr  # main.c:370-371
    r = run(["wm", "-m", "8", "-p_size", "10", "-n", "1000", "-alphabet", "5", "-c", "-dry"], tmp_path)
    assert r.returncode == 1  # wu_determine_shiftsize rejects the alphabet (wu/wu.c:45-46)
    r = run(["ac", "-m", "8", "-p_size", "10", "-n", "1000", "-alphabet", "4"], tmp_path)
    assert r.returncode == 1 and "does not exist (use -c" in r.stderr


@no_spawn
def test_create_then_dry_run_builds_the_reference_tables(tmp_path):
    n, m, p, sigma = 200000, 8, 100, 4
    args = ["all", "-m", str(m), "-p_size", str(p), "-n", str(n), "-alphabet", str(sigma)]
    r = run(args + ["-c", "-dry"], tmp_path)
    assert r.returncode == 0, r.stderr
    tfile = tmp_path / "data-cuda-multi" / "text" / ("text%d_%d" % (sigma, n))
    pfile = tmp_path / "data-cuda-multi" / "pattern" / str(n) / str(m) / str(sigma) / "pattern"
    text = np.fromfile(tfile, dtype=np.uint8)
    pat = np.fromfile(pfile, dtype=np.uint8)
    assert text.size == n and np.array_equal(text, O.gen_text(n, 42, sigma))  # the corpus of the tests and bench
    assert pat.size == m * p and pat.max() < sigma
    hits = sum(1 for j in range(0, p, 2) if text.tobytes().find(pat[j * m:(j + 1) * m].tobytes()) >= 0)
    assert hits == p // 2  # create_multiple_pattern_with_hits' role: every even pattern occurs in the text
    _, t = O.oracle_ac(pat, m, p, sigma)
    assert "preproc_ac states \t%d\t patterns \t%d\t" % (t.idcounter, t.patterncounter) in r.stdout
    ts = O.oracle_sh(pat, m, p, sigma)[1]
    assert "preproc_sh states \t%d\t patterns \t%d\t" % (ts.idcounter, ts.patterncounter) in r.stdout
    assert "text symbols \t%d\t fnv1a64 \t%016x\n" % (n, fnv1a64(text)) in r.stdout
    # second run: files exist, nothing is created, same tables
    r2 = run(args + ["-dry"], tmp_path)
    assert r2.returncode == 0 and "created" not in r2.stdout
    assert [ln for ln in r2.stdout.splitlines() if ln.startswith("text symbols")] == \
           [ln for ln in r.stdout.splitlines() if ln.startswith("text symbols")]
    if S.device_count() == 0:
        r3 = run(args, tmp_path)  # no GPU: the search fails loudly, there is no CPU fallback
        assert r3.returncode == 1 and "search_ac" in r3.stderr and "Total results" not in r3.stdout


@no_spawn
@pytest.mark.parametrize("coding,sigma,letters", [("dna", 4, "ACGT"), ("protein", 20, "ACDEFGHIKLMNPQRSTVWY")])
def test_fasta_decoding(tmp_path, coding, sigma, letters):
    n, m, p = 50000, 6, 50
    sym = O.gen_text(n, 99, sigma)
    fasta = tmp_path / "in.fa"
    write_fasta(fasta, sym, letters, junk="NNRY" if coding == "dna" else "XB*Z")  # not symbols of the coding
    r = run(["all", "-m", str(m), "-p_size", str(p), "-n", str(n), "-alphabet", str(sigma), "-text", str(fasta),
             "-coding", coding, "-pattern", str(tmp_path / "pat"), "-c", "-dry"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "text symbols \t%d\t fnv1a64 \t%016x\n" % (n, fnv1a64(sym)) in r.stdout
    # asking for more symbols than the file holds is an error, as is a coding / alphabet mismatch
    r = run(["ac", "-m", str(m), "-p_size", str(p), "-n", str(n + 1), "-alphabet", str(sigma), "-text", str(fasta),
             "-coding", coding, "-pattern", str(tmp_path / "pat2"), "-c", "-dry"], tmp_path)
    assert r.returncode == 1 and "fewer than -n" in r.stderr
    r = run(["ac", "-m", str(m), "-p_size", str(p), "-n", str(n), "-alphabet", "8", "-text", str(fasta),
             "-coding", coding, "-dry"], tmp_path)
    assert r.returncode == 1 and "you must use an alphabet size of" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,multi", [(1, False), (3, False), (1, True)])
def test_driver_totals_match_oracle_on_a_fasta_text(tmp_path, capfd, ranks, multi):
    if not os.path.exists(DSO):
        pytest.fail("smatcher_main.so missing: `make -C cuda-aho-corasick-wu-manber_amd` builds it")
    n, m, p, sigma = 1500007, 12, 400, 4
    sym = O.gen_text(n, 5, sigma)
    fasta = tmp_path / "genome.fa"
    write_fasta(fasta, sym, "ACGT")
    pfile = tmp_path / "pattern"
    drv = C.CDLL(DSO)
    args = [b"smatcher", b"all", b"-m", str(m).encode(), b"-p_size", str(p).encode(), b"-n", str(n).encode(),
            b"-alphabet", str(sigma).encode(), b"-text", str(fasta).encode(), b"-coding", b"dna",
            b"-pattern", str(pfile).encode(), b"-c", b"-ranks", str(ranks).encode()] + ([b"-multi"] if multi else [])
    argv = (C.c_char_p * len(args))(*args)
    capfd.readouterr()
    assert drv.smatcher_main(len(args), argv) == 0
    out = capfd.readouterr().out
    pat = np.fromfile(pfile, dtype=np.uint8)
    want, _ = O.oracle_ac(pat, m, p, sigma, sym)
    assert want >= p // 2
    assert "Total results (ac): %d.\n" % want in out and "Total results: %d.\n" % want in out
    # the side-by-side multi-device pass (smh_multi_*) runs for R > 1 when R devices are visible, or on request
    assert ("multi-device ac (%d devices" % ranks in out) == (multi or (ranks > 1 and S.device_count() >= ranks))
    if multi:
        assert "multi-device ac (1 devices, RCCL all-reduce) matches \t%d\t" % want in out
        assert "multi-device wm (1 devices, RCCL all-reduce) matches \t%d\t" % want in out
    assert "Total results (sh): %d.\n" % want in out and out.count("search_sh matches") == ranks
    assert "Total results (sbom): %d.\n" % want in out and out.count("search_sbom matches") == ranks
    assert out.count("search_ac matches") == ranks and out.count("search_wm2 matches") == ranks
    assert out.count("Kernel 5 matches") == 3 * ranks and "gpuTime[5]:" in out  # cuda_ac5, cuda_sh5, cuda_sbom5
    per_rank = [int(ln.split("\t")[1]) for ln in out.splitlines() if ln.startswith("search_ac matches")]
    shard = []
    for r in range(ranks):
        b, e = O.shard_range(n, ranks, r, m)
        shard.append(O.oracle_ac(pat, m, p, sigma, sym[b:e])[0])
    assert per_rank == shard  # each rank's count is the oracle's count of its byte range (main.c:467-477)
