"""ctypes binding of tests/emu/libsmh_emu.so: the kernels' lane code compiled for the CPU
(TEST HARNESS ONLY -- see tests/emu/emu_kernels.cpp)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-aho-corasick-wu-manber_amd")
sys.path.insert(0, PKG)
import smatcher_hip as S  # noqa: E402

_PATH = os.path.join(ROOT, "tests", "emu", "libsmh_emu.so")
if not os.path.exists(_PATH):
    subprocess.check_call(["make", "-s", "-C", PKG, "emu"])
_emu = C.CDLL(_PATH)
_emu.emu_ac_scan.restype = C.c_uint64
_emu.emu_ac_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_int, C.c_uint32]
_emu.emu_wm_scan.restype = C.c_uint64
_emu.emu_wm_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_int, C.c_uint32]


def ac_scan(ac, text, variant=S.VARIANT_TUNED, blocks=0):
    """Scan with whatever stride / depth plan the handle currently holds (AcAutomaton.set_scan_plan)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(_emu.emu_ac_scan(ac.h, text.ctypes.data_as(S.u8p), len(text), variant, blocks))


# ---- the parts of a handle's text-independent engine (csrc/ac_host.c): internal entry point, bound for the tests
S.lib.smh_ac_flat_part.restype = C.c_void_p
S.lib.smh_ac_flat_part.argtypes = [C.c_void_p, C.c_int]


def ac_scan_flat_parts(ac, text, blocks=0):
    """-> (sum over the parts of the emulated scan of `text`, number of parts): what SMH_ENGINE_AC_FLAT launches one after the other"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    total, i = 0, 0
    while True:
        part = S.lib.smh_ac_flat_part(ac.h, i)
        if not part:
            return total, i
        total += int(_emu.emu_ac_scan(C.c_void_p(part), text.ctypes.data_as(S.u8p), len(text), S.VARIANT_TUNED, blocks))
        i += 1


def wm_scan(wm, text, variant=S.VARIANT_TUNED, blocks=0):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(_emu.emu_wm_scan(wm.h, text.ctypes.data_as(S.u8p), len(text), variant, blocks))


_emu.emu_ac_positions.restype = C.c_uint64
_emu.emu_ac_positions.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32]
_emu.emu_wm_positions.restype = C.c_uint64
_emu.emu_wm_positions.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32]


def _positions(fn, handle, text, capacity, blocks):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    out = np.zeros(max(capacity, 1), dtype=np.uint64)
    total = int(fn(handle, text.ctypes.data_as(S.u8p), len(text), out.ctypes.data_as(C.POINTER(C.c_uint64)), capacity, blocks))
    return total, out[:min(total, capacity)]


def ac_positions(ac, text, capacity, blocks=0):
    return _positions(_emu.emu_ac_positions, ac.h, text, capacity, blocks)


def wm_positions(wm, text, capacity, blocks=0):
    return _positions(_emu.emu_wm_positions, wm.h, text, capacity, blocks)


_emu.emu_sh_scan.restype = C.c_uint64
_emu.emu_sh_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, S.i32p, C.c_int, C.c_uint32]


def sh_scan(sh, text, bmbc=None, variant=S.VARIANT_TUNED, blocks=0):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    b = None
    if bmbc is not None:
        bmbc = np.ascontiguousarray(bmbc, dtype=np.int32)
        b = bmbc.ctypes.data_as(S.i32p)
    return int(_emu.emu_sh_scan(sh.h, text.ctypes.data_as(S.u8p), len(text), b, variant, blocks))


_emu.emu_sbom_scan.restype = C.c_uint64
_emu.emu_sbom_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_int, C.c_uint32]


def sbom_scan(sb, text, variant=S.VARIANT_TUNED, blocks=0):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(_emu.emu_sbom_scan(sb.h, text.ctypes.data_as(S.u8p), len(text), variant, blocks))


_emu.emu_ac_positions_tuned.restype = C.c_uint64
_emu.emu_ac_positions_tuned.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32]


def ac_positions_tuned(ac, text, capacity, blocks=0):
    """Positions mode of the tuned scan kernels under the handle's current plan; total is None when the
    plan's halo exceeds what the mode covers (the runtime then uses the per-segment kernel)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    out = np.zeros(max(capacity, 1), dtype=np.uint64)
    total = int(_emu.emu_ac_positions_tuned(ac.h, text.ctypes.data_as(S.u8p), len(text),
                                            out.ctypes.data_as(C.POINTER(C.c_uint64)), capacity, blocks))
    if total == 0xFFFFFFFFFFFFFFFF:
        return None, out[:0]
    return total, out[:min(total, capacity)]


_emu.emu_wm_positions_tuned.restype = C.c_uint64
_emu.emu_wm_positions_tuned.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32]


def wm_positions_tuned(wm, text, capacity, blocks=0):
    return _positions(_emu.emu_wm_positions_tuned, wm.h, text, capacity, blocks)


_emu.emu_wm_scan_multi.restype = C.c_uint64
_emu.emu_wm_scan_multi.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, S.u8p, C.c_uint64, C.POINTER(C.c_uint64),
                                   C.c_uint64, C.c_uint32]


def wm_scan_multi(suffix, classes, text, capacity=None, blocks=0):
    """One-pass scan of a mixed-length set: `suffix` = WmTables over the patterns' last min-length symbols,
    `classes` = one WmTables per length (ascending).  capacity None: count; else (total, positions)."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    arr = (C.c_void_p * len(classes))(*[c.h for c in classes])
    if capacity is None:
        return int(_emu.emu_wm_scan_multi(suffix.h, arr, len(classes), text.ctypes.data_as(S.u8p), len(text), None, 0, blocks))
    out = np.zeros(max(capacity, 1), dtype=np.uint64)
    total = int(_emu.emu_wm_scan_multi(suffix.h, arr, len(classes), text.ctypes.data_as(S.u8p), len(text),
                                       out.ctypes.data_as(C.POINTER(C.c_uint64)), capacity, blocks))
    return total, out[:min(total, capacity)]


# ---- grouped pair-gram filter of a mixed-length set (csrc/wm_host.c): internal entry point, bound for the tests
def build_gram_mixed(suffix, patterns, lengths, lib=None):
    """attach the grouped pair-gram filter over the FULL patterns to the suffix handle; 0 = built, 1 = not applicable.
    lib: the library whose builder runs (the testing twin's obeys the development knobs); default the product's"""
    lib = lib or S.lib
    lib.smh_wm_build_gram_mixed.restype = C.c_int
    lib.smh_wm_build_gram_mixed.argtypes = [C.c_void_p, S.u8p, C.POINTER(C.c_uint32), C.c_int]
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    return int(lib.smh_wm_build_gram_mixed(suffix.h, patterns.ctypes.data_as(S.u8p),
                                           lengths.ctypes.data_as(C.POINTER(C.c_uint32)), len(lengths)))


# ---- mixed-length automaton (csrc/acm_host.c): internal entry points of the library, bound here for the tests
S.lib.smh_acm_compile.restype = C.c_void_p
S.lib.smh_acm_compile.argtypes = [S.u8p, C.POINTER(C.c_uint32), C.c_int, C.c_int]
S.lib.smh_acm_free.restype = None
S.lib.smh_acm_free.argtypes = [C.c_void_p]
_emu.emu_acm_scan.restype = C.c_uint64
_emu.emu_acm_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_uint32]


def acm_compile(patterns, lengths, sigma):
    """-> handle (int) or None when no cut of the automaton is worth one pass (message in smh_last_error)"""
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
    h = S.lib.smh_acm_compile(patterns.ctypes.data_as(S.u8p), lengths.ctypes.data_as(C.POINTER(C.c_uint32)), len(lengths), sigma)
    return h


def acm_scan(h, text, blocks=0):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(_emu.emu_acm_scan(h, text.ctypes.data_as(S.u8p), len(text), blocks))


_emu.emu_sog_scan.restype = C.c_uint64
_emu.emu_sog_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_uint32]


def sog_scan(sg, text, blocks=0):
    """the table-walking SOG lane code (sog_lane.h) over the handle's tables"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return int(_emu.emu_sog_scan(sg.h, text.ctypes.data_as(S.u8p), len(text), blocks))


_emu.emu_keys_scan.restype = C.c_uint64
_emu.emu_keys_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.c_uint32]
_emu.emu_keys_positions.restype = C.c_uint64
_emu.emu_keys_positions.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32]


def keys_scan(keys, text, blocks=0):
    """the key engine's lane code (csrc/key_lane.h) over `text`; `keys` = smatcher_hip.KeyTable"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    got = int(_emu.emu_keys_scan(keys.h, text.ctypes.data_as(S.u8p), len(text), blocks))
    assert got != 0xFFFFFFFFFFFFFFFF, "the two guard-page placements of the text disagree"
    return got


def keys_positions(keys, text, capacity, blocks=0):
    return _positions(_emu.emu_keys_positions, keys.h, text, capacity, blocks)


_emu.emu_hash_scan.restype = C.c_uint64
_emu.emu_hash_scan.argtypes = [C.c_void_p, S.u8p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]


def hash_scan(wm, text, blocks=0):
    """-> (count, columns that passed the filter) by the window-hash engine's lane code (csrc/hash_lane.h) of a WmTables handle"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    ev = C.c_uint64(0)
    got = int(_emu.emu_hash_scan(wm.h, text.ctypes.data_as(S.u8p), len(text), None, 0, blocks, C.byref(ev)))
    assert got != 0xFFFFFFFFFFFFFFFF, "the handle keeps no window-hash engine"
    assert got != 0xFFFFFFFFFFFFFFFE, "the two guard-page placements of the text disagree"
    return got, int(ev.value)


def hash_positions(wm, text, capacity, blocks=0):
    text = np.ascontiguousarray(text, dtype=np.uint8)
    out = np.zeros(max(capacity, 1), dtype=np.uint64)
    total = int(_emu.emu_hash_scan(wm.h, text.ctypes.data_as(S.u8p), len(text), out.ctypes.data_as(C.POINTER(C.c_uint64)), capacity, blocks, None))
    return total, out[:min(total, capacity)]
