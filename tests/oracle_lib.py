"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been
built, the reference's own compiled ac/ac.c + wu/wu.c (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
_REF = os.path.join(ORACLE_DIR, "_ref", "libref.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
i64p = C.POINTER(C.c_int64)


def build_oracle():
    """(Re)build liboracle.so, and libref.so when /root/reference is present."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


def _load():
    if not os.path.exists(_LIB):
        build_oracle()
    lib = C.CDLL(_LIB)
    lib.ora_splitmix64_at.restype = C.c_uint64
    lib.ora_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
    lib.ora_gen_text.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    lib.ora_gen_patterns_uniform.argtypes = [u8p, C.c_int, C.c_int, C.c_uint64, C.c_int]
    lib.ora_gen_patterns_mixed.argtypes = [u8p, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                           C.c_uint64, C.c_uint64, C.c_int]
    lib.ora_preproc_ac.restype = C.c_void_p
    lib.ora_preproc_ac.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p, u32p]
    lib.ora_ac_idcounter.restype = C.c_uint32
    lib.ora_ac_idcounter.argtypes = [C.c_void_p]
    lib.ora_ac_patterncounter.restype = C.c_uint32
    lib.ora_ac_patterncounter.argtypes = [C.c_void_p]
    lib.ora_search_ac.restype = C.c_uint64
    lib.ora_search_ac.argtypes = [u8p, C.c_int64, C.c_void_p]
    lib.ora_search_ac_tables.restype = C.c_uint64
    lib.ora_search_ac_tables.argtypes = [u8p, C.c_int64, C.c_int, i32p, u32p, u32p]
    lib.ora_free_ac.argtypes = [C.c_void_p]
    lib.ora_preproc_sh.restype = None
    lib.ora_preproc_sh.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p, u32p, u32p]
    lib.ora_pre_bmbc.restype = None
    lib.ora_pre_bmbc.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p]
    lib.ora_search_sh.restype = C.c_uint64
    lib.ora_search_sh.argtypes = [C.c_int, u8p, C.c_int64, C.c_int, i32p, u32p, i32p]
    lib.ora_preproc_sbom.restype = None
    lib.ora_preproc_sbom.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p, u32p, u32p]
    lib.ora_search_sbom.restype = C.c_uint64
    lib.ora_search_sbom.argtypes = [u8p, C.c_int, u8p, C.c_int64, C.c_int, i32p, u32p]
    lib.ora_wu_determine_shiftsize.restype = C.c_uint32
    lib.ora_wu_determine_shiftsize.argtypes = [C.c_int]
    wu_tabs = [i32p, i32p, i32p, i32p]
    lib.ora_preproc_wu.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + wu_tabs
    lib.ora_preproc_wu2.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + wu_tabs
    lib.ora_search_wu.restype = C.c_uint64
    lib.ora_search_wu.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, u8p, C.c_int64, C.c_int] + wu_tabs
    lib.ora_search_wu2.restype = C.c_uint64
    lib.ora_search_wu2.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int64, C.c_int] + wu_tabs
    lib.ora_preproc_wu_csr.restype = None
    lib.ora_preproc_wu_csr.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, i32p, u32p, i32p, i32p]
    lib.ora_search_wu_csr.restype = C.c_uint64
    lib.ora_search_wu_csr.argtypes = [u8p, C.c_int, u8p, C.c_int64, C.c_int, i32p, u32p, i32p, i32p]
    lib.ora_preproc_sog8.restype = None
    lib.ora_preproc_sog8.argtypes = [u8p, u32p, i32p, u8p, C.POINTER(u8p), C.c_int]
    lib.ora_search_sog8.restype = C.c_uint64
    lib.ora_search_sog8.argtypes = [u8p, u32p, i32p, u8p, C.POINTER(u8p), u8p, C.c_int64, C.c_int]
    lib.ora_shard_range.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, i64p, i64p]
    lib.ora_count_bruteforce.restype = C.c_uint64
    lib.ora_count_bruteforce.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int64]
    lib.ora_positions_bruteforce.restype = C.c_uint64
    lib.ora_positions_bruteforce.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int64, i64p, C.c_uint64]
    lib.ora_fnv1a64.restype = C.c_uint64
    lib.ora_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
    return lib


lib = _load()

NBITS = 2  # m_nBitsInShift, main.c:431
B = 3      # main.c:335


# ------------------------------------------------------------------ corpus
def gen_text(n, seed=42, sigma=4, offset=0):
    out = np.empty(n, dtype=np.uint8)
    lib.ora_gen_text(_ptr(out, u8p), n, offset, seed, sigma)
    return out


def gen_patterns(m, p, seed=7, sigma=4):
    out = np.empty(m * p, dtype=np.uint8)
    lib.ora_gen_patterns_uniform(_ptr(out, u8p), m, p, seed, sigma)
    return out


def gen_patterns_mixed(m, p, seed=7, sigma=4, text_seed=42, n_text=0, every=2):
    out = np.empty(m * p, dtype=np.uint8)
    lib.ora_gen_patterns_mixed(_ptr(out, u8p), m, p, seed, sigma, text_seed, n_text, every)
    return out


def _pattern_ptrs(pat_flat, m, p, pad=1):
    """char** view with each row copied to its own (m+pad)-byte zero padded buffer."""
    rows = np.zeros((p, m + pad), dtype=np.uint8)
    rows[:, :m] = np.asarray(pat_flat, dtype=np.uint8).reshape(p, m)
    arr = (u8p * p)()
    for j in range(p):
        arr[j] = C.cast(rows[j].ctypes.data, u8p)
    return arr, rows  # keep rows alive


def fnv(a):
    a = np.ascontiguousarray(a)
    return int(lib.ora_fnv1a64(a.ctypes.data, a.nbytes))


# ------------------------------------------------------------------ AC
class ACTables:
    """Flat tables exactly as main.c:410-420 allocates and initialises them."""

    def __init__(self, m, p, sigma):
        rows = m * p + 1
        self.m, self.p, self.sigma, self.rows = m, p, sigma, rows
        self.state_transition = np.full(rows * sigma, -1, dtype=np.int32)
        self.state_supply = np.zeros(rows, dtype=np.uint32)
        self.state_final = np.zeros(rows, dtype=np.uint32)
        self.idcounter = 0
        self.patterncounter = 0

    def ptrs(self):
        return (_ptr(self.state_transition, i32p), _ptr(self.state_supply, u32p),
                _ptr(self.state_final, u32p))


def oracle_ac(pat_flat, m, p, sigma, text=None):
    """-> (count or None, ACTables) from the restatement (oracle/ora_ac.c)."""
    t = ACTables(m, p, sigma)
    arr, keep = _pattern_ptrs(pat_flat, m, p)
    h = lib.ora_preproc_ac(arr, m, p, sigma, *t.ptrs())
    t.idcounter = lib.ora_ac_idcounter(h)
    t.patterncounter = lib.ora_ac_patterncounter(h)
    count = None
    if text is not None:
        text = np.ascontiguousarray(text, dtype=np.uint8)
        count = int(lib.ora_search_ac(_ptr(text, u8p), len(text), h))
        count2 = int(lib.ora_search_ac_tables(_ptr(text, u8p), len(text), sigma, *t.ptrs()))
        assert count == count2, "oracle trie walk and flat-table walk disagree"
    lib.ora_free_ac(h)
    del keep
    return count, t


def oracle_ac_search_tables(text, sigma, t):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(lib.ora_search_ac_tables(_ptr(text, u8p), len(text), sigma, *t.ptrs()))


# ------------------------------------------------------------------ WM
class WMTables:
    """Dense tables exactly as main.c:429-449 allocates and initialises them.
    PREFIX_value / PREFIX_index are left uninitialised by the reference; zeros here."""

    def __init__(self, m, p, sigma, shiftsize):
        self.m, self.p, self.sigma, self.shiftsize = m, p, sigma, shiftsize
        self.SHIFT = np.full(shiftsize, m - B + 1, dtype=np.int32)
        self.PREFIX_value = np.zeros(shiftsize * p, dtype=np.int32)
        self.PREFIX_index = np.zeros(shiftsize * p, dtype=np.int32)
        self.PREFIX_size = np.zeros(shiftsize, dtype=np.int32)

    def ptrs(self):
        return (_ptr(self.SHIFT, i32p), _ptr(self.PREFIX_value, i32p),
                _ptr(self.PREFIX_index, i32p), _ptr(self.PREFIX_size, i32p))

    def digest(self):
        """Digest of the defined part of the tables (bucket entries below PREFIX_size)."""
        vals, idxs = [], []
        p = self.p
        for h in np.nonzero(self.PREFIX_size)[0]:
            k = int(self.PREFIX_size[h])
            vals.append(self.PREFIX_value[h * p:h * p + k])
            idxs.append(self.PREFIX_index[h * p:h * p + k])
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int32)
        return (fnv(self.SHIFT), fnv(self.PREFIX_size), fnv(cat(vals)), fnv(cat(idxs)))


def oracle_wu(pat_flat, m, p, sigma, text=None, flat=True):
    shiftsize = int(lib.ora_wu_determine_shiftsize(sigma))
    if shiftsize == 0:
        raise ValueError("The alphabet size is not supported by wu-manber")
    t = WMTables(m, p, sigma, shiftsize)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    count = None
    if flat:
        lib.ora_preproc_wu2(_ptr(pat_flat, u8p), m, p, sigma, B, NBITS, *t.ptrs())
        if text is not None:
            text = np.ascontiguousarray(text, dtype=np.uint8)
            count = int(lib.ora_search_wu2(_ptr(pat_flat, u8p), m, p, _ptr(text, u8p), len(text),
                                           NBITS, *t.ptrs()))
    else:
        arr, keep = _pattern_ptrs(pat_flat, m, p, pad=0)
        lib.ora_preproc_wu(arr, m, p, sigma, B, NBITS, *t.ptrs())
        if text is not None:
            text = np.ascontiguousarray(text, dtype=np.uint8)
            count = int(lib.ora_search_wu(arr, m, p, _ptr(text, u8p), len(text), NBITS, *t.ptrs()))
        del keep
    return count, t


class WMTablesCSR:
    """The same tables in compressed rows (oracle/ora_wu.c, CSR form): what makes alphabet 256 / 100 000
    patterns checkable -- the dense arrays would be 2 x 2.1 GB."""

    def __init__(self, pat_flat, m, p, sigma):
        self.m, self.p, self.sigma = m, p, sigma
        self.shiftsize = int(lib.ora_wu_determine_shiftsize(sigma))
        if self.shiftsize == 0:
            raise ValueError("The alphabet size is not supported by wu-manber")
        self.pat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
        self.SHIFT = np.full(self.shiftsize, m - B + 1, dtype=np.int32)
        self.bucket_off = np.zeros(self.shiftsize + 1, dtype=np.uint32)
        self.bucket_val = np.zeros(max(p, 1), dtype=np.int32)
        self.bucket_idx = np.zeros(max(p, 1), dtype=np.int32)
        lib.ora_preproc_wu_csr(_ptr(self.pat, u8p), m, p, B, NBITS, self.shiftsize, _ptr(self.SHIFT, i32p),
                               _ptr(self.bucket_off, u32p), _ptr(self.bucket_val, i32p), _ptr(self.bucket_idx, i32p))

    def search(self, text):
        """search_wu2 over the compressed rows; re-entrant (ctypes releases the GIL)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        return int(lib.ora_search_wu_csr(_ptr(self.pat, u8p), self.m, _ptr(text, u8p), len(text), NBITS,
                                         _ptr(self.SHIFT, i32p), _ptr(self.bucket_off, u32p),
                                         _ptr(self.bucket_val, i32p), _ptr(self.bucket_idx, i32p)))

    def digest(self):
        """Same four digests as WMTables.digest(): SHIFT, PREFIX_size, defined PREFIX_value / PREFIX_index."""
        sizes = np.diff(self.bucket_off.astype(np.int64)).astype(np.int32)
        return (fnv(self.SHIFT), fnv(sizes), fnv(self.bucket_val[:self.p]), fnv(self.bucket_idx[:self.p]))


# ------------------------------------------------------------------ SOG (sog/sog8.c)
class SogTables:
    """Caller-owned tables as main.c:495-515 allocates them."""

    def __init__(self, p):
        self.p = p
        self.T8 = np.zeros(1 << 24, dtype=np.uint8)
        self.scanner_hs = np.zeros(p, dtype=np.uint32)
        self.scanner_index = np.zeros(p, dtype=np.int32)
        self.scanner_hs2 = np.zeros(32 * 256, dtype=np.uint8)

    def ptrs(self):
        return (_ptr(self.T8, u8p), _ptr(self.scanner_hs, u32p), _ptr(self.scanner_index, i32p), _ptr(self.scanner_hs2, u8p))


def oracle_sog8(pat_flat, p, text=None):
    """-> (count or None, SogTables) from the restatement (2-level bitmap from the real hash)"""
    t = SogTables(p)
    arr, keep = _pattern_ptrs(pat_flat, 8, p, pad=0)
    lib.ora_preproc_sog8(*t.ptrs(), arr, p)
    cnt = None
    if text is not None:
        text = np.ascontiguousarray(text, dtype=np.uint8)
        cnt = int(lib.ora_search_sog8(*t.ptrs(), arr, _ptr(text, u8p), len(text), p))
    del keep
    return cnt, t


def ref_sog8(pat_flat, p, text, tabs):
    """the reference's own preproc_sog8 + search_sog8 (count undefined: see ora_sog.c); fills `tabs`"""
    r = ref()
    r.ref_run_sog8.restype = C.c_ulonglong
    r.ref_run_sog8.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, u32p, i32p, u8p]
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(r.ref_run_sog8(_ptr(pat_flat, u8p), p, _ptr(text, u8p), len(text), *tabs.ptrs()))


# ------------------------------------------------------------------ misc
def shard_range(n, R, i, m):
    b, e = C.c_int64(), C.c_int64()
    lib.ora_shard_range(n, R, i, m, C.byref(b), C.byref(e))
    return b.value, e.value


def count_bruteforce(pat_flat, m, p, text):
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(lib.ora_count_bruteforce(_ptr(pat_flat, u8p), m, p, _ptr(text, u8p), len(text)))


def positions_bruteforce(pat_flat, m, p, text):
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    total = int(lib.ora_positions_bruteforce(_ptr(pat_flat, u8p), m, p, _ptr(text, u8p), len(text),
                                             None, 0))
    out = np.empty(total, dtype=np.int64)
    if total:
        lib.ora_positions_bruteforce(_ptr(pat_flat, u8p), m, p, _ptr(text, u8p), len(text),
                                     _ptr(out, i64p), total)
    return out


# ------------------------------------------------------------------ compiled reference
# ------------------------------------------------------------------ Set-Horspool
class SHTables:
    """Flat tables as main.c:410-420,427 allocates them for multish: the reversed trie and bmBc."""

    def __init__(self, m, p, sigma):
        rows = m * p + 1
        self.m, self.p, self.sigma, self.rows = m, p, sigma, rows
        self.state_transition = np.full(rows * sigma, -1, dtype=np.int32)
        self.state_final = np.zeros(rows, dtype=np.uint32)
        self.bmBc = np.zeros(sigma, dtype=np.int32)
        self.idcounter = 0
        self.patterncounter = 0


def pre_bmbc(pat_flat, m, p, sigma):
    """The set-Horspool bad-character table the reference's missing helper computes (oracle/ora_sh.c)."""
    out = np.zeros(sigma, dtype=np.int32)
    arr, keep = _pattern_ptrs(pat_flat, m, p)
    lib.ora_pre_bmbc(arr, m, p, sigma, _ptr(out, i32p))
    del keep
    return out


def oracle_sh(pat_flat, m, p, sigma, text=None):
    """-> (count or None, SHTables) from the restatement (oracle/ora_sh.c)."""
    t = SHTables(m, p, sigma)
    arr, keep = _pattern_ptrs(pat_flat, m, p)
    idc, pc = C.c_uint32(), C.c_uint32()
    lib.ora_preproc_sh(arr, m, p, sigma, _ptr(t.state_transition, i32p), _ptr(t.state_final, u32p),
                       C.byref(idc), C.byref(pc))
    lib.ora_pre_bmbc(arr, m, p, sigma, _ptr(t.bmBc, i32p))
    t.idcounter, t.patterncounter = idc.value, pc.value
    count = None
    if text is not None:
        text = np.ascontiguousarray(text, dtype=np.uint8)
        count = int(lib.ora_search_sh(m, _ptr(text, u8p), len(text), sigma, _ptr(t.state_transition, i32p),
                                      _ptr(t.state_final, u32p), _ptr(t.bmBc, i32p)))
    del keep
    return count, t


def ref_sh(pat_flat, m, p, sigma, text=None, bmBc=None):
    """-> (count, SHTables) from the reference's own sh/sh.c; bmBc defaults to the oracle's table."""
    t = SHTables(m, p, sigma)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    t.bmBc = np.ascontiguousarray(bmBc if bmBc is not None else pre_bmbc(pat_flat, m, p, sigma), dtype=np.int32)
    idc, pc = C.c_uint32(), C.c_uint32()
    tp, ts = C.c_double(), C.c_double()
    tptr = _ptr(np.ascontiguousarray(text, dtype=np.uint8), u8p) if text is not None else None
    n = len(text) if text is not None else 0
    cnt = ref().ref_run_sh(_ptr(pat_flat, u8p), m, p, sigma, tptr, n, _ptr(t.state_transition, i32p),
                           _ptr(t.state_final, u32p), _ptr(t.bmBc, i32p), C.byref(idc), C.byref(pc),
                           C.byref(tp), C.byref(ts))
    t.idcounter, t.patterncounter = idc.value, pc.value
    return (int(cnt) if text is not None else None), t


# ------------------------------------------------------------------ SBOM
class SBOMTables:
    """Flat tables as main.c:410-425 allocates them for multisbom: the factor oracle and, per state,
    {count, pattern ids...} in rows of 200 entries."""

    def __init__(self, m, p, sigma):
        rows = m * p + 1
        self.m, self.p, self.sigma, self.rows = m, p, sigma, rows
        self.state_transition = np.full(rows * sigma, -1, dtype=np.int32)
        self.state_final_multi = np.zeros(rows * 200, dtype=np.uint32)
        self.idcounter = 0
        self.patterncounter = 0


def oracle_sbom(pat_flat, m, p, sigma, text=None):
    """-> (count or None, SBOMTables) from the restatement (oracle/ora_sbom.c)."""
    t = SBOMTables(m, p, sigma)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    arr, keep = _pattern_ptrs(pat_flat, m, p)
    idc, pc = C.c_uint32(), C.c_uint32()
    lib.ora_preproc_sbom(arr, m, p, sigma, _ptr(t.state_transition, i32p), _ptr(t.state_final_multi, u32p),
                         C.byref(idc), C.byref(pc))
    t.idcounter, t.patterncounter = idc.value, pc.value
    count = None
    if text is not None:
        text = np.ascontiguousarray(text, dtype=np.uint8)
        count = int(lib.ora_search_sbom(_ptr(pat_flat, u8p), m, _ptr(text, u8p), len(text), sigma,
                                        _ptr(t.state_transition, i32p), _ptr(t.state_final_multi, u32p)))
    del keep
    return count, t


def ref_sbom(pat_flat, m, p, sigma, text=None):
    """-> (count, SBOMTables) from the reference's own sbom/sbom.c."""
    t = SBOMTables(m, p, sigma)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    idc, pc = C.c_uint32(), C.c_uint32()
    tptr = _ptr(np.ascontiguousarray(text, dtype=np.uint8), u8p) if text is not None else None
    n = len(text) if text is not None else 0
    cnt = ref().ref_run_sbom(_ptr(pat_flat, u8p), m, p, sigma, tptr, n, _ptr(t.state_transition, i32p),
                             _ptr(t.state_final_multi, u32p), C.byref(idc), C.byref(pc), None, None)
    t.idcounter, t.patterncounter = idc.value, pc.value
    return (int(cnt) if text is not None else None), t


def have_ref():
    return os.path.exists(_REF)


_ref = None


def ref():
    global _ref
    if _ref is None:
        r = C.CDLL(_REF)
        r.ref_shiftsize.restype = C.c_uint
        r.ref_shiftsize.argtypes = [C.c_int]
        r.ref_run_ac.restype = C.c_ulonglong
        r.ref_run_ac.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, i32p, u32p, u32p,
                                 u32p, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        r.ref_run_sh.restype = C.c_ulonglong
        r.ref_run_sh.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, i32p, u32p, i32p,
                                 u32p, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        r.ref_run_sbom.restype = C.c_ulonglong
        r.ref_run_sbom.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, i32p, u32p,
                                   u32p, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        r.ref_run_wu.restype = C.c_ulonglong
        r.ref_run_wu.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, i32p, i32p, i32p,
                                 i32p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _ref = r
    return _ref


def ref_ac(pat_flat, m, p, sigma, text=None):
    """-> (count, ACTables, t_preproc, t_search) from the reference's own ac/ac.c."""
    t = ACTables(m, p, sigma)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    idc, pc = C.c_uint32(), C.c_uint32()
    tp, ts = C.c_double(), C.c_double()
    tptr = _ptr(np.ascontiguousarray(text, dtype=np.uint8), u8p) if text is not None else None
    n = len(text) if text is not None else 0
    cnt = ref().ref_run_ac(_ptr(pat_flat, u8p), m, p, sigma, tptr, n, *t.ptrs(),
                           C.byref(idc), C.byref(pc), C.byref(tp), C.byref(ts))
    t.idcounter, t.patterncounter = idc.value, pc.value
    return (int(cnt) if text is not None else None), t, tp.value, ts.value


def ref_wu(pat_flat, m, p, sigma, text=None, flat=True):
    shiftsize = int(ref().ref_shiftsize(sigma))
    t = WMTables(m, p, sigma, shiftsize)
    pat_flat = np.ascontiguousarray(pat_flat, dtype=np.uint8)
    tp, ts = C.c_double(), C.c_double()
    tptr = _ptr(np.ascontiguousarray(text, dtype=np.uint8), u8p) if text is not None else None
    n = len(text) if text is not None else 0
    cnt = ref().ref_run_wu(_ptr(pat_flat, u8p), m, p, sigma, tptr, n, *t.ptrs(),
                           1 if flat else 0, C.byref(tp), C.byref(ts))
    return (int(cnt) if text is not None else None), t, tp.value, ts.value
