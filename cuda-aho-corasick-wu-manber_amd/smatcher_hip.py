"""ctypes mirror of libsmatcher_hip.so (include/smatcher.h + include/smatcher_hip.h).

This is harness plumbing for tests/ and bench.py: it adds nothing to the hot
path, which is C host code + gfx950 kernels inside the shared library.  The
library is built in-tree by `make -C cuda-aho-corasick-wu-manber_amd` (or
__graft_entry__.build()); a missing library is a hard error -- there is no
Python or CPU fallback for the search entry points.

Processes that also use torch must import torch BEFORE this module: torch loads its bundled HIP
runtime by path, this library requests libamdhip64.so.7 by SONAME and then shares torch's copy;
the other order puts two HIP runtimes into one process and torch then finds no GPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# the product library -- unless this file is executed by load_testing() below, which presets LIB_PATH
LIB_PATH = globals().get("LIB_PATH") or os.path.join(HERE, "libsmatcher_hip.so")
TESTING_LIB_PATH = os.path.join(os.path.dirname(HERE), "tests", "emu", "libsmatcher_hip_testing.so")
IS_TESTING_LIB = os.path.basename(LIB_PATH) != "libsmatcher_hip.so"
TUNE_WM, TUNE_AC, TUNE_HASH, TUNE_KEY, TUNE_PSET = range(5)  # csrc/smh_tune.h

SMH_OK = 0
VARIANT_TUNED = 0
VARIANT_TABLE = 1
ALGO_AC = 0
ALGO_WM = 1
ENGINE_AC_FLAT = 2
ENGINE_KEYS = 3
ENGINE_HASH = 4
ENGINES = 5
ENGINE_NAMES = {0: "automaton kernels", 1: "suffix-filter kernels", 2: "plain stride-1 automata", 3: "key table", 4: "window-hash filter"}

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int)
u32p = C.POINTER(C.c_uint)
u64p = C.POINTER(C.c_uint64)
dblp = C.POINTER(C.c_double)


class SmhError(RuntimeError):
    pass


class AcInfo(C.Structure):
    _fields_ = [("alphabet", C.c_uint32), ("m", C.c_uint32), ("states", C.c_uint32),
                ("finals", C.c_uint32), ("rows", C.c_uint32), ("entry_bytes", C.c_uint32),
                ("lds_rows", C.c_uint32), ("lds_bytes", C.c_uint32), ("table_bytes", C.c_uint64),
                ("scan_depth", C.c_uint32), ("scan_stride", C.c_uint32), ("scan_exact", C.c_uint32),
                ("scan_full_rows", C.c_uint32), ("scan_engine", C.c_uint32), ("scan_dense", C.c_uint32),
                ("verify_in_registers", C.c_uint32), ("gram_kind", C.c_uint32), ("adaptive", C.c_uint32),
                ("flat_parts", C.c_uint32), ("key_slots", C.c_uint32), ("hash_slots", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class WmInfo(C.Structure):
    _fields_ = [("alphabet", C.c_uint32), ("m", C.c_uint32), ("patterns", C.c_uint32),
                ("distinct", C.c_uint32), ("shiftsize", C.c_uint32), ("shift_zero", C.c_uint32),
                ("block_symbols", C.c_uint32), ("filter_log2", C.c_uint32),
                ("filter_exact", C.c_uint32), ("filter_hashed", C.c_uint32),
                ("verify_slots", C.c_uint32), ("lds_bytes", C.c_uint32), ("scan_engine", C.c_uint32),
                ("gram_planes", C.c_uint32), ("verify_in_registers", C.c_uint32), ("gram_kind", C.c_uint32),
                ("adaptive", C.c_uint32), ("key_slots", C.c_uint32), ("hash_slots", C.c_uint32), ("verify_ck_slots", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class AdaptInfo(C.Structure):
    """smh_adapt_info: what the adaptive engine knows about the text on the current device"""
    _fields_ = [("struct_size", C.c_uint32), ("adaptive", C.c_uint32), ("engine", C.c_uint32), ("flips", C.c_uint32),
                ("reports", C.c_uint32), ("reserved", C.c_uint32), ("ms_per_gib", C.c_double * ENGINES),
                ("events_per_4k", C.c_double * ENGINES), ("est_ms_per_gib", C.c_double * ENGINES), ("verify_density", C.c_double)]


class KeysInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("alphabet", C.c_uint32), ("m", C.c_uint32), ("keys", C.c_uint32),
                ("key_bits", C.c_uint32), ("slot_bytes", C.c_uint32), ("slots", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("est_ms_per_gib", C.c_double), ("layout", C.c_uint32), ("overflow_keys", C.c_uint32)]


class PsetInfo(C.Structure):
    _fields_ = [("alphabet", C.c_uint32), ("algorithm", C.c_uint32), ("classes", C.c_uint32),
                ("patterns", C.c_uint32), ("min_length", C.c_uint32), ("max_length", C.c_uint32),
                ("one_pass", C.c_uint32), ("passes", C.c_uint32)]


class ShInfo(C.Structure):
    _fields_ = [("alphabet", C.c_uint32), ("m", C.c_uint32), ("states", C.c_uint32), ("finals", C.c_uint32),
                ("tuned_engine", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class SbomInfo(C.Structure):
    _fields_ = [("alphabet", C.c_uint32), ("m", C.c_uint32), ("states", C.c_uint32), ("patterns", C.c_uint32),
                ("listed", C.c_uint32), ("tuned_engine", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class SbomTable(C.Structure):
    """struct sbom_table (include/smatcher.h)"""
    _fields_ = [("idcounter", C.c_uint), ("patterncounter", C.c_uint), ("zerostate", C.c_void_p)]


class AcTable(C.Structure):
    """struct ac_table (include/smatcher.h)"""
    _fields_ = [("idcounter", C.c_uint), ("patterncounter", C.c_uint), ("zerostate", C.c_void_p)]


# every symbol include/smatcher.h and include/smatcher_hip.h declare
LEGACY_SYMBOLS = (["fail", "preproc_ac", "search_ac", "free_ac", "wu_determine_shiftsize",
                   "preproc_wu", "preproc_wu2", "search_wu", "search_wu2", "m_nBitsInShift",
                   "shiftsize"]
                  + ["cuda_ac%d" % k for k in range(1, 6)] + ["cuda_wm%d" % k for k in range(1, 6)]
                  + ["preBmBc", "preproc_sh", "search_sh", "free_sh"] + ["cuda_sh%d" % k for k in range(1, 6)]
                  + ["preproc_sbom", "search_sbom", "free_sbom", "pointer_array"] + ["cuda_sbom%d" % k for k in range(1, 6)]
                  + ["preproc_sog8", "search_sog8"] + ["cuda_sog%d" % k for k in range(1, 6)])
EXT_SYMBOLS = ["smh_version", "smh_last_error", "smh_device_count", "smh_set_device",
               "smh_device_name", "smh_device_pci_bus_id", "smh_device_malloc", "smh_device_free", "smh_device_memset",
               "smh_copy_to_device", "smh_copy_to_host", "smh_stream_synchronize", "smh_stream_read_probe",
               "smh_stream_read_probe_variant", "smh_host_path_release", "smh_host_path_set_piece", "smh_legacy_handle_builds",
               "smh_splitmix64_at", "smh_corpus_text_host", "smh_corpus_text_device",
               "smh_corpus_patterns", "smh_corpus_text_host_kind", "smh_corpus_text_device_kind", "smh_corpus_patterns_kind",
               "smh_shard_range", "smh_ac_compile_tables",
               "smh_ac_compile_patterns", "smh_ac_get_info", "smh_ac_get_adapt", "smh_wm_get_adapt", "smh_ac_set_scan_plan", "smh_ac_set_scan_engine", "smh_ac_positions", "smh_wm_positions", "smh_ac_scan", "smh_ac_count_host",
               "smh_ac_free", "smh_wm_compile", "smh_wm_compile_tables", "smh_wm_get_info", "smh_wm_set_scan_engine",
               "smh_wm_scan", "smh_wm_count_host", "smh_wm_free", "smh_pset_compile", "smh_pset_get_info",
               "smh_pset_get_class", "smh_pset_scan", "smh_pset_positions", "smh_pset_count_host",
               "smh_pset_free", "smh_keys_compile_patterns", "smh_keys_get_info", "smh_keys_scan", "smh_keys_positions", "smh_keys_free",
               "smh_sh_compile_tables", "smh_sh_compile_patterns", "smh_sh_get_info",
               "smh_sh_valid_bmbc", "smh_sh_scan", "smh_sh_count_host", "smh_sh_free", "smh_sbom_compile_tables",
               "smh_sbom_compile_patterns", "smh_sbom_get_info", "smh_sbom_scan", "smh_sbom_count_host", "smh_sbom_free",
               "smh_sog_compile_tables", "smh_sog_scan", "smh_sog_count_host", "smh_sog_free",
               "smh_multi_create", "smh_multi_device_count", "smh_multi_uses_rccl", "smh_multi_load_text",
               "smh_multi_generate_text", "smh_multi_ac_count", "smh_multi_wm_count", "smh_multi_ac_prepare",
               "smh_multi_wm_prepare", "smh_multi_free"]


def _load():
    if not os.path.exists(LIB_PATH):
        raise SmhError("libsmatcher_hip.so is not built (%s); run `make -C %s` -- there is no "
                       "fallback path" % (LIB_PATH, HERE))
    lib = C.CDLL(LIB_PATH)
    lib.smh_version.restype = C.c_char_p
    lib.smh_last_error.restype = C.c_char_p
    lib.smh_device_count.restype = C.c_int
    lib.smh_set_device.argtypes = [C.c_int]
    lib.smh_device_name.argtypes = [C.c_char_p, C.c_size_t]
    lib.smh_device_pci_bus_id.argtypes = [C.c_char_p, C.c_size_t]
    lib.smh_device_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_uint64]
    lib.smh_device_free.argtypes = [C.c_void_p]
    lib.smh_device_memset.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
    lib.smh_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.smh_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.smh_stream_synchronize.argtypes = [C.c_void_p]
    lib.smh_stream_read_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.smh_stream_read_probe_variant.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    lib.smh_host_path_release.restype = None
    lib.smh_host_path_release.argtypes = []
    lib.smh_legacy_handle_builds.restype = C.c_uint64
    lib.smh_legacy_handle_builds.argtypes = []
    lib.smh_host_path_set_piece.restype = C.c_uint64
    lib.smh_host_path_set_piece.argtypes = [C.c_uint64]
    lib.smh_splitmix64_at.restype = C.c_uint64
    lib.smh_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
    lib.smh_corpus_text_host.restype = None
    lib.smh_corpus_text_host.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    lib.smh_corpus_text_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    lib.smh_corpus_patterns.restype = None
    lib.smh_corpus_patterns.argtypes = [u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    lib.smh_corpus_text_host_kind.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    lib.smh_corpus_text_device_kind.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    lib.smh_corpus_patterns_kind.argtypes = [u8p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    lib.smh_shard_range.restype = None
    lib.smh_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, u64p]
    lib.smh_ac_compile_tables.restype = C.c_void_p
    lib.smh_ac_compile_tables.argtypes = [i32p, u32p, u32p, C.c_uint64, C.c_int, C.c_int]
    lib.smh_ac_compile_patterns.restype = C.c_void_p
    lib.smh_ac_compile_patterns.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
    lib.smh_ac_get_info.argtypes = [C.c_void_p, C.POINTER(AcInfo)]
    lib.smh_ac_get_adapt.argtypes = [C.c_void_p, C.POINTER(AdaptInfo)]
    lib.smh_wm_get_adapt.argtypes = [C.c_void_p, C.POINTER(AdaptInfo)]
    lib.smh_ac_set_scan_plan.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.smh_ac_set_scan_engine.argtypes = [C.c_void_p, C.c_int]
    lib.smh_ac_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    lib.smh_ac_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.smh_wm_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.smh_ac_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, u64p, dblp]
    lib.smh_ac_free.restype = None
    lib.smh_ac_free.argtypes = [C.c_void_p]
    lib.smh_wm_compile.restype = C.c_void_p
    lib.smh_wm_compile.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
    lib.smh_wm_compile_tables.restype = C.c_void_p
    lib.smh_wm_compile_tables.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
    lib.smh_wm_get_info.argtypes = [C.c_void_p, C.POINTER(WmInfo)]
    lib.smh_wm_set_scan_engine.argtypes = [C.c_void_p, C.c_int]
    lib.smh_wm_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    lib.smh_wm_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, u64p, dblp]
    lib.smh_wm_free.restype = None
    lib.smh_wm_free.argtypes = [C.c_void_p]
    lib.smh_pset_compile.restype = C.c_void_p
    lib.smh_pset_compile.argtypes = [u8p, u32p, C.c_int, C.c_int, C.c_int]
    lib.smh_pset_get_info.argtypes = [C.c_void_p, C.POINTER(PsetInfo)]
    lib.smh_pset_get_class.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p]
    lib.smh_pset_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.smh_pset_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.smh_pset_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, u64p, dblp]
    lib.smh_pset_free.restype = None
    lib.smh_pset_free.argtypes = [C.c_void_p]
    lib.smh_sh_compile_tables.restype = C.c_void_p
    lib.smh_sh_compile_tables.argtypes = [i32p, u32p, C.c_uint64, C.c_int, C.c_int]
    lib.smh_sh_compile_patterns.restype = C.c_void_p
    lib.smh_sh_compile_patterns.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
    lib.smh_sh_get_info.argtypes = [C.c_void_p, C.POINTER(ShInfo)]
    lib.smh_sh_valid_bmbc.argtypes = [C.c_void_p, i32p]
    lib.smh_sh_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, i32p, C.c_void_p, C.c_int, C.c_void_p]
    lib.smh_sh_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, i32p, C.c_int, u64p, dblp]
    lib.smh_sh_free.restype = None
    lib.smh_sh_free.argtypes = [C.c_void_p]
    lib.smh_sbom_compile_tables.restype = C.c_void_p
    lib.smh_sbom_compile_tables.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, u32p, C.c_uint64]
    lib.smh_sbom_compile_patterns.restype = C.c_void_p
    lib.smh_sbom_compile_patterns.argtypes = [u8p, C.c_int, C.c_int, C.c_int]
    lib.smh_sbom_get_info.argtypes = [C.c_void_p, C.POINTER(SbomInfo)]
    lib.smh_sbom_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    lib.smh_sbom_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, u64p, dblp]
    lib.smh_sbom_free.restype = None
    lib.smh_sbom_free.argtypes = [C.c_void_p]
    lib.preproc_sbom.restype = C.POINTER(SbomTable)
    lib.preproc_sbom.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p]
    lib.search_sbom.restype = C.c_uint
    lib.search_sbom.argtypes = [C.POINTER(u8p), C.c_int, u8p, C.c_int, C.POINTER(SbomTable)]
    lib.free_sbom.restype = None
    lib.free_sbom.argtypes = [C.POINTER(SbomTable), C.c_int]
    for k in range(1, 6):
        f = getattr(lib, "cuda_sbom%d" % k)
        f.restype = None
        f.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, i32p, u32p]
    lib.preBmBc.restype = None
    lib.preBmBc.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p]
    lib.preproc_sh.restype = C.POINTER(AcTable)
    lib.preproc_sh.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p]
    lib.search_sh.restype = C.c_uint
    lib.search_sh.argtypes = [C.c_int, u8p, C.c_int, C.POINTER(AcTable), i32p]
    lib.free_sh.restype = None
    lib.free_sh.argtypes = [C.POINTER(AcTable), C.c_int]
    for k in range(1, 6):
        f = getattr(lib, "cuda_sh%d" % k)
        f.restype = None
        f.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.c_int, i32p, u32p, i32p]
    # legacy names, with the reference's argument lists (smatcher.h)
    lib.preproc_ac.restype = C.POINTER(AcTable)
    lib.preproc_ac.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, i32p, u32p, u32p]
    lib.search_ac.restype = C.c_uint
    lib.search_ac.argtypes = [u8p, C.c_int, C.POINTER(AcTable)]
    lib.free_ac.restype = None
    lib.free_ac.argtypes = [C.POINTER(AcTable), C.c_int]
    lib.wu_determine_shiftsize.restype = None
    lib.wu_determine_shiftsize.argtypes = [C.c_int]
    tabs = [i32p, i32p, i32p, i32p]
    lib.preproc_wu.restype = None
    lib.preproc_wu.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int] + tabs
    lib.preproc_wu2.restype = None
    lib.preproc_wu2.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int] + tabs
    lib.search_wu.restype = C.c_uint
    lib.search_wu.argtypes = [C.POINTER(u8p), C.c_int, C.c_int, u8p, C.c_int] + tabs
    lib.search_wu2.restype = C.c_uint
    lib.search_wu2.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int] + tabs
    for k in range(1, 6):
        f = getattr(lib, "cuda_ac%d" % k)
        f.restype = None
        f.argtypes = [C.c_int, u8p, C.c_int, C.c_int, C.c_int, i32p, u32p, u32p]
        g = getattr(lib, "cuda_wm%d" % k)
        g.restype = C.c_int
        g.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int] + tabs + [dblp]
    lib.preproc_sog8.restype = None
    lib.preproc_sog8.argtypes = [u8p, u32p, i32p, u8p, C.POINTER(u8p), C.c_int, u8p, C.c_int, C.c_int, C.c_int]
    lib.search_sog8.restype = C.c_uint
    lib.search_sog8.argtypes = [u8p, u32p, i32p, u8p, C.POINTER(u8p), C.c_int, u8p, C.c_int, C.c_int, C.c_int]
    for k in range(1, 6):
        f = getattr(lib, "cuda_sog%d" % k)
        f.restype = None
        f.argtypes = [u8p, u32p, i32p, u8p, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]
    lib.smh_sog_compile_tables.restype = C.c_void_p
    lib.smh_sog_compile_tables.argtypes = [u8p, u32p, i32p, u8p, u8p, C.c_int]
    lib.smh_sog_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    lib.smh_sog_count_host.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), dblp]
    lib.smh_sog_free.restype = None
    lib.smh_sog_free.argtypes = [C.c_void_p]
    lib.smh_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.smh_multi_device_count.argtypes = [C.c_void_p]
    lib.smh_multi_uses_rccl.argtypes = [C.c_void_p]
    lib.smh_multi_load_text.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_int]
    lib.smh_multi_generate_text.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    lib.smh_multi_ac_count.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), dblp]
    lib.smh_multi_wm_count.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), dblp]
    lib.smh_multi_ac_prepare.argtypes = [C.c_void_p, C.c_void_p]
    lib.smh_multi_wm_prepare.argtypes = [C.c_void_p, C.c_void_p]
    lib.smh_multi_free.restype = None
    lib.smh_multi_free.argtypes = [C.c_void_p]
    return lib


lib = _load()


def load_testing():
    """A second copy of this module bound to tests/emu/libsmatcher_hip_testing.so: the product's sources built with -DSMH_TESTING,
    the only build that has the development knobs (csrc/smh_tune.h) and the runtime's test hooks.  Tests and tools that force a
    code path or run a timing experiment use it (`T = S.load_testing(); T.tune(T.TUNE_WM, "gram=6")`); everything else -- and
    every number on a bench record -- goes through the product library, which reads no knob."""
    import importlib.util
    import sys
    name = __name__ + "_testing"
    if name in sys.modules:
        return sys.modules[name]
    if not os.path.exists(TESTING_LIB_PATH):
        raise SmhError("%s is not built; run `make -C %s emu`" % (TESTING_LIB_PATH, HERE))
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    mod.LIB_PATH = TESTING_LIB_PATH
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


def for_tools():
    """development micro-drivers (tools/): the testing twin when a knob variable is exported, the product library otherwise"""
    import sys
    if any(os.environ.get(v) for v in ("SMH_WM_TUNE", "SMH_AC_TUNE", "SMH_HASH_TUNE", "SMH_KEY_TUNE", "SMH_PSET_TUNE")):
        print("[tools] development knob exported: running tests/emu/libsmatcher_hip_testing.so, not the product library", file=sys.stderr)
        return load_testing()
    return sys.modules[__name__]


def tune(which, text=None):
    """testing library only: set (or clear, text=None) one knob string -- smh_test_tune_set, csrc/smh_tune.c"""
    if not IS_TESTING_LIB:
        raise SmhError("the product library has no development knobs; use load_testing().tune(...)")
    lib.smh_test_tune_set.argtypes = [C.c_int, C.c_char_p]
    if lib.smh_test_tune_set(which, text.encode() if text else None) != 0:
        raise SmhError("smh_test_tune_set(%r): no such knob class" % (which,))


def tune_clear():
    for which in range(5):
        tune(which, None)


def _check(rc, what):
    if rc != SMH_OK:
        raise SmhError("%s failed (%d): %s" % (what, rc, lib.smh_last_error().decode()))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def shiftsize_global():
    return C.c_uint.in_dll(lib, "shiftsize").value


def device_count():
    return int(lib.smh_device_count())


def device_name():
    buf = C.create_string_buffer(256)
    _check(lib.smh_device_name(buf, 256), "smh_device_name")
    return buf.value.decode()


def device_pci_bus_id():
    """domain:bus:device.function of the current device"""
    buf = C.create_string_buffer(64)
    _check(lib.smh_device_pci_bus_id(buf, 64), "smh_device_pci_bus_id")
    return buf.value.decode()


CORPUS_UNIFORM, CORPUS_DNA_REPEATS, CORPUS_SKEWED, CORPUS_PLANTED = 0, 1, 2, 3
CORPUS_NAMES = {CORPUS_UNIFORM: "uniform", CORPUS_DNA_REPEATS: "dna_repeats", CORPUS_SKEWED: "skewed", CORPUS_PLANTED: "planted"}


def corpus_text(n, seed=42, alphabet=4, offset=0, kind=CORPUS_UNIFORM):
    out = np.empty(n, dtype=np.uint8)
    _check(lib.smh_corpus_text_host_kind(out.ctypes.data_as(u8p), n, offset, seed, alphabet, kind), "smh_corpus_text_host_kind")
    return out


def corpus_text_device(d_ptr, n, seed=42, alphabet=4, offset=0, kind=CORPUS_UNIFORM, stream=None):
    _check(lib.smh_corpus_text_device_kind(C.c_void_p(d_ptr), n, offset, seed, alphabet, kind, C.c_void_p(stream or 0)),
           "smh_corpus_text_device_kind")


def corpus_patterns(m, p, seed=7, alphabet=4, text_seed=42, n_text=0, every=2, kind=CORPUS_UNIFORM):
    out = np.empty(m * p, dtype=np.uint8)
    _check(lib.smh_corpus_patterns_kind(out.ctypes.data_as(u8p), m, p, seed, alphabet, text_seed, n_text, every, kind),
           "smh_corpus_patterns_kind")
    return out


def shard_range(n, shards, i, m):
    b, e = C.c_uint64(), C.c_uint64()
    lib.smh_shard_range(n, shards, i, m, C.byref(b), C.byref(e))
    return b.value, e.value


class AcAutomaton:
    """smh_ac handle: compiled Aho-Corasick automaton."""

    def __init__(self, handle):
        if not handle:
            raise SmhError("AC compile failed: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def from_patterns(cls, pat_flat, m, p, alphabet):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_ac_compile_patterns(ptr, m, p, alphabet))

    @classmethod
    def from_tables(cls, state_transition, state_supply, state_final, rows, alphabet, m):
        return cls(lib.smh_ac_compile_tables(state_transition.ctypes.data_as(i32p),
                                             state_supply.ctypes.data_as(u32p),
                                             state_final.ctypes.data_as(u32p), rows, alphabet, m))

    def info(self):
        out = AcInfo()
        _check(lib.smh_ac_get_info(self.h, C.byref(out)), "smh_ac_get_info")
        return out

    def adapt(self):
        out = AdaptInfo(struct_size=C.sizeof(AdaptInfo))
        _check(lib.smh_ac_get_adapt(self.h, C.byref(out)), "smh_ac_get_adapt")
        return out

    def set_scan_plan(self, stride=0, depth=0):
        _check(lib.smh_ac_set_scan_plan(self.h, stride, depth), "smh_ac_set_scan_plan")

    def set_scan_engine(self, engine):
        _check(lib.smh_ac_set_scan_engine(self.h, engine), "smh_ac_set_scan_engine")

    def scan_device(self, d_text_ptr, n, d_count_ptr, variant=VARIANT_TUNED, stream=None):
        _check(lib.smh_ac_scan(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_count_ptr), variant,
                               C.c_void_p(stream or 0)), "smh_ac_scan")

    def positions_device(self, d_text_ptr, n, d_positions_ptr, capacity, d_cursor_ptr, stream=None):
        _check(lib.smh_ac_positions(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_positions_ptr), capacity,
                                    C.c_void_p(d_cursor_ptr), C.c_void_p(stream or 0)), "smh_ac_positions")

    def count_host(self, text, variant=VARIANT_TUNED):
        t, ptr = _u8(text)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_ac_count_host(self.h, ptr, len(t), variant, C.byref(cnt), C.byref(secs)),
               "smh_ac_count_host")
        return cnt.value, secs.value

    def close(self):
        if self.h:
            lib.smh_ac_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class WmTables:
    """smh_wm handle: compiled Wu-Manber tables."""

    def __init__(self, handle):
        if not handle:
            raise SmhError("WM compile failed: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def from_patterns(cls, pat_flat, m, p, alphabet):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_wm_compile(ptr, m, p, alphabet))

    @classmethod
    def from_tables(cls, pat_flat, m, p, alphabet, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_wm_compile_tables(ptr, m, p, alphabet, SHIFT.ctypes.data_as(i32p),
                                             PREFIX_value.ctypes.data_as(i32p),
                                             PREFIX_index.ctypes.data_as(i32p),
                                             PREFIX_size.ctypes.data_as(i32p)))

    def info(self):
        out = WmInfo()
        _check(lib.smh_wm_get_info(self.h, C.byref(out)), "smh_wm_get_info")
        return out

    def adapt(self):
        out = AdaptInfo(struct_size=C.sizeof(AdaptInfo))
        _check(lib.smh_wm_get_adapt(self.h, C.byref(out)), "smh_wm_get_adapt")
        return out

    def set_scan_engine(self, engine):
        _check(lib.smh_wm_set_scan_engine(self.h, engine), "smh_wm_set_scan_engine")

    def scan_device(self, d_text_ptr, n, d_count_ptr, variant=VARIANT_TUNED, stream=None):
        _check(lib.smh_wm_scan(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_count_ptr), variant,
                               C.c_void_p(stream or 0)), "smh_wm_scan")

    def positions_device(self, d_text_ptr, n, d_positions_ptr, capacity, d_cursor_ptr, stream=None):
        _check(lib.smh_wm_positions(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_positions_ptr), capacity,
                                    C.c_void_p(d_cursor_ptr), C.c_void_p(stream or 0)), "smh_wm_positions")

    def count_host(self, text, variant=VARIANT_TUNED):
        t, ptr = _u8(text)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_wm_count_host(self.h, ptr, len(t), variant, C.byref(cnt), C.byref(secs)),
               "smh_wm_count_host")
        return cnt.value, secs.value

    def close(self):
        if self.h:
            lib.smh_wm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KeyTable:
    """smh_keys handle: the key engine by itself (one exact hash-set lookup per text column)."""

    def __init__(self, pat_flat, m, p, alphabet):
        a, ptr = _u8(pat_flat)
        lib.smh_keys_compile_patterns.restype = C.c_void_p
        h = lib.smh_keys_compile_patterns(ptr, m, p, alphabet)
        if not h:
            raise SmhError("key engine: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(h)

    def info(self):
        out = KeysInfo(struct_size=C.sizeof(KeysInfo))
        _check(lib.smh_keys_get_info(self.h, C.byref(out)), "smh_keys_get_info")
        return out

    def scan_device(self, d_text_ptr, n, d_count_ptr, stream=None):
        _check(lib.smh_keys_scan(self.h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_void_p(d_count_ptr), C.c_void_p(stream or 0)), "smh_keys_scan")

    def positions_device(self, d_text_ptr, n, d_positions_ptr, capacity, d_cursor_ptr, stream=None):
        _check(lib.smh_keys_positions(self.h, C.c_void_p(d_text_ptr), C.c_uint64(n), C.c_void_p(d_positions_ptr), C.c_uint64(capacity),
                                      C.c_void_p(d_cursor_ptr), C.c_void_p(stream or 0)), "smh_keys_positions")

    def close(self):
        if self.h:
            lib.smh_keys_free.argtypes = [C.c_void_p]
            lib.smh_keys_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PatternSet:
    """smh_pset handle: patterns of mixed lengths, one compiled class per distinct length."""

    def __init__(self, patterns, lengths, alphabet, algorithm=ALGO_AC):
        a, ptr = _u8(patterns)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        if int(ln.sum()) != len(a):
            raise SmhError("PatternSet: lengths sum to %d, %d pattern bytes given" % (int(ln.sum()), len(a)))
        h = lib.smh_pset_compile(ptr, ln.ctypes.data_as(u32p), len(ln), alphabet, algorithm)
        if not h:
            raise SmhError("pattern set compile failed: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(h)

    def info(self):
        out = PsetInfo()
        _check(lib.smh_pset_get_info(self.h, C.byref(out)), "smh_pset_get_info")
        return out

    def classes(self):
        out = []
        for i in range(self.info().classes):
            ln, cnt = C.c_uint32(), C.c_uint32()
            _check(lib.smh_pset_get_class(self.h, i, C.byref(ln), C.byref(cnt)), "smh_pset_get_class")
            out.append((ln.value, cnt.value))
        return out

    def scan_device(self, d_text_ptr, n, d_count_ptr, stream=None):
        _check(lib.smh_pset_scan(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_count_ptr),
                                 C.c_void_p(stream or 0)), "smh_pset_scan")

    def positions_device(self, d_text_ptr, n, d_positions_ptr, capacity, d_cursor_ptr, stream=None):
        _check(lib.smh_pset_positions(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_positions_ptr), capacity,
                                      C.c_void_p(d_cursor_ptr), C.c_void_p(stream or 0)), "smh_pset_positions")

    def count_host(self, text):
        t, ptr = _u8(text)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_pset_count_host(self.h, ptr, len(t), C.byref(cnt), C.byref(secs)), "smh_pset_count_host")
        return cnt.value, secs.value

    def close(self):
        if self.h:
            lib.smh_pset_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SogTables:
    """SOG (sog/sog8.c): the caller-owned tables filled by preproc_sog8, and a handle over them."""

    def __init__(self, pat_flat, p):
        self.p = p
        self.pat, _ = _u8(pat_flat)
        self.T8 = np.zeros(1 << 24, dtype=np.uint8)
        self.scanner_hs = np.zeros(p, dtype=np.uint32)
        self.scanner_index = np.zeros(p, dtype=np.int32)
        self.scanner_hs2 = np.zeros(32 * 256, dtype=np.uint8)
        rows = self.pat.reshape(p, 8)
        self._rows = [np.ascontiguousarray(r) for r in rows]
        self.ptrs = (u8p * p)(*[r.ctypes.data_as(u8p) for r in self._rows])
        lib.preproc_sog8(*self.tables(), self.ptrs, 8, None, 0, p, 3)
        self.h = lib.smh_sog_compile_tables(*self.tables(), self.pat.ctypes.data_as(u8p), p)
        if not self.h:
            raise SmhError("smh_sog_compile_tables: " + lib.smh_last_error().decode())

    def tables(self):
        return (self.T8.ctypes.data_as(u8p), self.scanner_hs.ctypes.data_as(u32p), self.scanner_index.ctypes.data_as(i32p),
                self.scanner_hs2.ctypes.data_as(u8p))

    def scan_device(self, d_text_ptr, n, d_count_ptr, variant=VARIANT_TUNED, stream=None):
        _check(lib.smh_sog_scan(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_count_ptr), variant, C.c_void_p(stream or 0)),
               "smh_sog_scan")

    def count_host(self, text, variant=VARIANT_TUNED):
        a, p = _u8(text)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_sog_count_host(self.h, p, len(a), variant, C.byref(cnt), C.byref(secs)), "smh_sog_count_host")
        return int(cnt.value), secs.value

    def close(self):
        if self.h:
            lib.smh_sog_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MULTI_HOST_SUM, MULTI_NO_RCCL, MULTI_SHARE_DEVICE = 1, 2, 4


class MultiGpu:
    """smh_multi_*: one process, several GPUs -- byte-range shards resident per device, counts added with one RCCL
    all-reduce (the reference driver's MPI_Scatterv / MPI_Reduce, main.c:464-489, 654-657)."""

    def __init__(self, n_devices, devices=None, flags=0):
        self.h = C.c_void_p()
        arr = (C.c_int * n_devices)(*devices) if devices is not None else None
        _check(lib.smh_multi_create(C.byref(self.h), arr, n_devices, flags), "smh_multi_create")

    @property
    def devices(self):
        return int(lib.smh_multi_device_count(self.h))

    @property
    def uses_rccl(self):
        return bool(lib.smh_multi_uses_rccl(self.h))

    def load_text(self, text, halo):
        a, p = _u8(text)
        _check(lib.smh_multi_load_text(self.h, p, len(a), halo), "smh_multi_load_text")

    def generate_text(self, n_total, seed, alphabet, halo):
        _check(lib.smh_multi_generate_text(self.h, n_total, seed, alphabet, halo), "smh_multi_generate_text")

    def _count(self, fn, handle, what):
        total, secs = C.c_uint64(), C.c_double()
        per = (C.c_uint64 * self.devices)()
        _check(fn(self.h, handle.h, C.byref(total), per, C.byref(secs)), what)
        return int(total.value), [int(x) for x in per], secs.value

    def ac_count(self, ac):
        return self._count(lib.smh_multi_ac_count, ac, "smh_multi_ac_count")

    def wm_count(self, wm):
        return self._count(lib.smh_multi_wm_count, wm, "smh_multi_wm_count")

    def prepare(self, handle):
        """table sets + one warm-up launch on every device, side by side (smh_multi_*_prepare)"""
        if isinstance(handle, AcAutomaton):
            _check(lib.smh_multi_ac_prepare(self.h, handle.h), "smh_multi_ac_prepare")
        else:
            _check(lib.smh_multi_wm_prepare(self.h, handle.h), "smh_multi_wm_prepare")

    def close(self):
        if self.h:
            lib.smh_multi_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShTrie:
    """smh_sh handle: Set-Horspool reversed trie + tuned engine."""

    def __init__(self, handle):
        if not handle:
            raise SmhError("SH compile failed: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def from_patterns(cls, pat_flat, m, p, alphabet):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_sh_compile_patterns(ptr, m, p, alphabet))

    @classmethod
    def from_tables(cls, state_transition, state_final, rows, alphabet, m):
        return cls(lib.smh_sh_compile_tables(state_transition.ctypes.data_as(i32p), state_final.ctypes.data_as(u32p),
                                             rows, alphabet, m))

    def info(self):
        out = ShInfo()
        _check(lib.smh_sh_get_info(self.h, C.byref(out)), "smh_sh_get_info")
        return out

    def valid_bmbc(self):
        out = np.zeros(self.info().alphabet, dtype=np.int32)
        _check(lib.smh_sh_valid_bmbc(self.h, out.ctypes.data_as(i32p)), "smh_sh_valid_bmbc")
        return out

    @staticmethod
    def _bm(bmbc):
        if bmbc is None:
            return None, None
        b = np.ascontiguousarray(bmbc, dtype=np.int32)
        return b, b.ctypes.data_as(i32p)

    def scan_device(self, d_text_ptr, n, d_count_ptr, bmbc=None, variant=VARIANT_TUNED, stream=None):
        keep, b = self._bm(bmbc)
        _check(lib.smh_sh_scan(self.h, C.c_void_p(d_text_ptr), n, b, C.c_void_p(d_count_ptr), variant,
                               C.c_void_p(stream or 0)), "smh_sh_scan")

    def count_host(self, text, bmbc=None, variant=VARIANT_TUNED):
        t, ptr = _u8(text)
        keep, b = self._bm(bmbc)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_sh_count_host(self.h, ptr, len(t), b, variant, C.byref(cnt), C.byref(secs)), "smh_sh_count_host")
        return cnt.value, secs.value

    def close(self):
        if self.h:
            lib.smh_sh_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SbomOracle:
    """smh_sbom handle: Set Backward Oracle Matching tables + tuned engine."""

    def __init__(self, handle):
        if not handle:
            raise SmhError("SBOM compile failed: %s" % lib.smh_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def from_patterns(cls, pat_flat, m, p, alphabet):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_sbom_compile_patterns(ptr, m, p, alphabet))

    @classmethod
    def from_tables(cls, pat_flat, m, p, alphabet, state_transition, state_final_multi, rows):
        a, ptr = _u8(pat_flat)
        return cls(lib.smh_sbom_compile_tables(ptr, m, p, alphabet, state_transition.ctypes.data_as(i32p),
                                               state_final_multi.ctypes.data_as(u32p), rows))

    def info(self):
        out = SbomInfo()
        _check(lib.smh_sbom_get_info(self.h, C.byref(out)), "smh_sbom_get_info")
        return out

    def scan_device(self, d_text_ptr, n, d_count_ptr, variant=VARIANT_TUNED, stream=None):
        _check(lib.smh_sbom_scan(self.h, C.c_void_p(d_text_ptr), n, C.c_void_p(d_count_ptr), variant,
                                 C.c_void_p(stream or 0)), "smh_sbom_scan")

    def count_host(self, text, variant=VARIANT_TUNED):
        t, ptr = _u8(text)
        cnt, secs = C.c_uint64(), C.c_double()
        _check(lib.smh_sbom_count_host(self.h, ptr, len(t), variant, C.byref(cnt), C.byref(secs)), "smh_sbom_count_host")
        return cnt.value, secs.value

    def close(self):
        if self.h:
            lib.smh_sbom_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
