"""Python face of the CPU oracle -- TEST INFRASTRUCTURE ONLY (see __init__.py).

* ``flagstat_python`` -- pure-Python loop, statement-for-statement restatement of
  ``libflagstats.h:118-142`` (small inputs only).
* ``flagstat_numpy``  -- vectorised numpy restatement of the same rule.
* ``flagstat_c`` / ``flagstat_hist`` / ``flagstat_mt`` / ``flagstat_generated`` --
  ctypes over ``oracle/liboracle.so`` (``flagstat_oracle.c``).
* ``load_ref`` / ``ref_call`` -- ctypes over ``oracle/_ref/libflagstats_ref.so``,
  the reference's own kernels compiled from ``/root/reference`` (may be absent).
* ``pyflagstats_dict`` -- restatement of the dict ``python/libflagstats.pyx:24-35``
  builds from the 32 counters.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

GEN_UNIFORM, GEN_NA12878, GEN_RAMP = 0, 1, 2

# python/libflagstats.pyx:24
SAM_FLAG_NAMES = [
    "FPAIRED", "FPROPER_PAIR", "FUNMAP", "FMUNMAP", "FREVERSE", "FMREVERSE",
    "FREAD1", "FREAD2", "FSECONDARY", "FQCFAIL", "FDUP", "FSUPPLEMENTARY",
    "n_pair_good", "n_sgltn", "n_pair_map",
]

# the 19 slots FLAGSTAT_scalar_update writes (libflagstats.h:118-142;
# benchmark/inmemory.cpp:173-194 lists 20 of which slot 9 is never non-zero)
LIVE_SLOTS = (2, 6, 7, 8, 10, 11, 12, 13, 14, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30)


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    target = ["all"] if ref else [os.path.join(_HERE, "liboracle.so")]
    subprocess.run(["make", "-s", "-C", _HERE] + target, check=True)


# --------------------------------------------------------------------------- #
# pure Python / numpy restatements
# --------------------------------------------------------------------------- #
def flagstat_python(values) -> np.ndarray:
    """libflagstats.h:118-142 as a per-flag Python loop (small inputs only)."""
    out = [0] * 32
    for v in values:
        v = int(v)
        w = 16 if (v & 512) else 0                       # :122
        if w:
            out[w + 9] += 1                              # :127
        if v & 256:
            out[w + 8] += 1                              # :129
        elif v & 2048:
            out[w + 11] += 1                             # :130
        elif v & 1:                                      # :131
            if (v & 2) and not (v & 4):
                out[w + 12] += 1                         # :133
            if v & 64:
                out[w + 6] += 1                          # :134
            if v & 128:
                out[w + 7] += 1                          # :135
            if (v & 8) and not (v & 4):
                out[w + 13] += 1                         # :136
            if not (v & 4) and not (v & 8):
                out[w + 14] += 1                         # :137
        if v & 4:
            out[w + 2] += 1                              # :140
        if v & 1024:
            out[w + 10] += 1                             # :141
    return np.asarray(out, dtype=np.uint64)


def flagstat_numpy(values: np.ndarray) -> np.ndarray:
    """Vectorised restatement of libflagstats.h:118-142 -> uint64[32]."""
    x = np.ascontiguousarray(values, dtype=np.uint16).ravel()
    out = np.zeros(32, dtype=np.uint64)
    if x.size == 0:
        return out
    bit = lambda m: (x & np.uint16(m)) != 0  # noqa: E731
    qc = bit(512)
    sec = bit(256)
    sup = bit(2048) & ~sec
    pp = bit(1) & ~sec & ~bit(2048)
    unm, mun = bit(4), bit(8)
    per_class = {
        2: unm,
        6: pp & bit(64),
        7: pp & bit(128),
        8: sec,
        10: bit(1024),
        11: sup,
        12: pp & bit(2) & ~unm,
        13: pp & mun & ~unm,
        14: pp & ~mun & ~unm,
    }
    for slot, m in per_class.items():
        out[slot] = np.count_nonzero(m & ~qc)
        out[16 + slot] = np.count_nonzero(m & qc)
    out[25] = np.count_nonzero(qc)
    return out


def pyflagstats_dict(counters, n_values: int) -> dict:
    """Restatement of python/libflagstats.pyx:24-35 (dict built from 32 counters)."""
    flags = np.asarray(counters)
    ret = {
        "n_values": n_values,
        "passed": dict(zip(SAM_FLAG_NAMES, flags[0:15])),
        "failed": dict(zip(SAM_FLAG_NAMES, flags[16:31])),
    }
    ret["passed"]["mapped"] = n_values - ret["passed"]["FUNMAP"] - ret["failed"]["FUNMAP"]
    ret["passed"]["paired_in_seq"] = ret["passed"]["FREAD1"] + ret["passed"]["FREAD2"]
    return ret


# --------------------------------------------------------------------------- #
# C restatement via ctypes
# --------------------------------------------------------------------------- #
_c = None
_ref = None
_U16P = ctypes.POINTER(ctypes.c_uint16)
_U32P = ctypes.POINTER(ctypes.c_uint32)
_U64P = ctypes.POINTER(ctypes.c_uint64)


def load_c():
    global _c
    if _c is not None:
        return _c
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build(ref=False)
    lib = ctypes.CDLL(path)
    lib.oracle_flagstat_u16.argtypes = [_U16P, ctypes.c_uint64, _U64P]
    lib.oracle_flagstat_u16.restype = None
    lib.oracle_flagstat_hist_u16.argtypes = [_U16P, ctypes.c_uint64, _U64P]
    lib.oracle_flagstat_hist_u16.restype = None
    lib.oracle_flagstat_mt_u16.argtypes = [_U16P, ctypes.c_uint64, ctypes.c_int, _U64P]
    lib.oracle_flagstat_mt_u16.restype = None
    lib.oracle_FLAGSTAT_scalar.argtypes = [_U16P, ctypes.c_uint32, _U32P]
    lib.oracle_FLAGSTAT_scalar.restype = ctypes.c_int
    lib.oracle_samtools_u16.argtypes = [_U16P, ctypes.c_uint64, _U64P]
    lib.oracle_samtools_u16.restype = None
    lib.oracle_pospopcnt_u16.argtypes = [_U16P, ctypes.c_uint64, _U64P]
    lib.oracle_pospopcnt_u16.restype = None
    lib.oracle_generate_u16.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32,
                                        ctypes.c_uint64, ctypes.c_uint64, _U16P]
    lib.oracle_generate_u16.restype = None
    lib.oracle_flagstat_generated.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32,
                                              ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, _U64P]
    lib.oracle_flagstat_generated.restype = None
    _c = lib
    return lib


def _as_u16(values) -> np.ndarray:
    a = np.asarray(values)
    if a.dtype != np.uint16:
        raise TypeError("oracle expects uint16")
    return a.ravel()  # keeps odd alignments of contiguous views


def _ptr16(a: np.ndarray):
    return ctypes.cast(a.ctypes.data, _U16P)


def _run(fn, values, *extra) -> np.ndarray:
    a = _as_u16(values)
    if not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a)
    out = np.zeros(32, dtype=np.uint64)
    fn(_ptr16(a), ctypes.c_uint64(a.size), *extra, out.ctypes.data_as(_U64P))
    return out


def flagstat_c(values) -> np.ndarray:
    """Branchy scalar loop (flagstat_oracle.c: oracle_flagstat_u16)."""
    return _run(load_c().oracle_flagstat_u16, values)


def flagstat_hist(values) -> np.ndarray:
    """Histogram evaluation, exactly equal, ~20x faster."""
    return _run(load_c().oracle_flagstat_hist_u16, values)


def flagstat_mt(values, threads: int | None = None) -> np.ndarray:
    threads = threads or (os.cpu_count() or 1)
    return _run(load_c().oracle_flagstat_mt_u16, values, ctypes.c_int(threads))


SAMTOOLS_FIELDS = ["n_reads", "n_mapped", "n_pair_all", "n_pair_map", "n_pair_good", "n_sgltn", "n_read1", "n_read2",
                   "n_dup", "n_diffchr", "n_diffhigh", "n_secondary", "n_supp"]  # bam_flagstat_t, benchmark/flagstats.cpp:42-48


def samtools_counts(values) -> dict:
    """The reference bench's samtools loop (benchmark/flagstats.cpp:51-70): {field: [pass, fail]}."""
    a = _as_u16(values)
    if not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a)
    out = np.zeros(26, dtype=np.uint64)
    load_c().oracle_samtools_u16(_ptr16(a), a.size, out.ctypes.data_as(_U64P))
    return {f: [int(out[2 * i]), int(out[2 * i + 1])] for i, f in enumerate(SAMTOOLS_FIELDS)}


def samtools_counts_python(values) -> dict:
    """Pure-Python statement of the same loop (small inputs): pins the C one."""
    out = {f: [0, 0] for f in SAMTOOLS_FIELDS}
    for c in (int(v) for v in np.asarray(values).ravel()):
        w = 1 if c & 512 else 0
        out["n_reads"][w] += 1
        if c & 256:
            out["n_secondary"][w] += 1
        elif c & 2048:
            out["n_supp"][w] += 1
        elif c & 1:
            out["n_pair_all"][w] += 1
            if (c & 2) and not (c & 4):
                out["n_pair_good"][w] += 1
            if c & 64:
                out["n_read1"][w] += 1
            if c & 128:
                out["n_read2"][w] += 1
            if (c & 8) and not (c & 4):
                out["n_sgltn"][w] += 1
            if not (c & 4) and not (c & 8):
                out["n_pair_map"][w] += 1
        if not (c & 4):
            out["n_mapped"][w] += 1
        if c & 1024:
            out["n_dup"][w] += 1
    return out


def samtools_text(counts: dict) -> str:
    """The report the reference prints from that struct (benchmark/flagstats.cpp:73-78 `percent`, :577-588)."""
    def pct(n, total):
        return "N/A" if total == 0 else "%.2f%%" % (float(np.float32(n) / np.float32(total)) * 100.0)

    s = counts
    two = lambda k: "%d + %d" % (s[k][0], s[k][1])  # noqa: E731
    par = lambda a, b: "(%s : %s)" % (pct(s[a][0], s[b][0]), pct(s[a][1], s[b][1]))  # noqa: E731
    return "".join(line + "\n" for line in [
        two("n_reads") + " in total (QC-passed reads + QC-failed reads)",
        two("n_secondary") + " secondary",
        two("n_supp") + " supplementary",
        two("n_dup") + " duplicates",
        two("n_mapped") + " mapped " + par("n_mapped", "n_reads"),
        two("n_pair_all") + " paired in sequencing",
        two("n_read1") + " read1",
        two("n_read2") + " read2",
        two("n_pair_good") + " properly paired " + par("n_pair_good", "n_pair_all"),
        two("n_pair_map") + " with itself and mate mapped",
        two("n_sgltn") + " singletons " + par("n_sgltn", "n_pair_all"),
    ])


def pospopcnt(values) -> np.ndarray:
    """uint64[16] positional popcount (flagstat_oracle.c: oracle_pospopcnt_u16)."""
    a = _as_u16(values)
    out = np.zeros(16, dtype=np.uint64)
    load_c().oracle_pospopcnt_u16(_ptr16(a), a.size, out.ctypes.data_as(_U64P))
    return out


def pospopcnt_numpy(values) -> np.ndarray:
    x = np.ascontiguousarray(values, dtype=np.uint16).ravel()
    return np.array([np.count_nonzero(x & np.uint16(1 << j)) for j in range(16)], dtype=np.uint64)


def ref_pospopcnt(values, naive: bool = False):
    """The reference's STORM_pospopcnt_u16 (or its scalar_naive form) from oracle/_ref, or None."""
    lib = load_ref()
    name = "ref_STORM_pospopcnt_u16_scalar_naive" if naive else "ref_STORM_pospopcnt_u16"
    if lib is None or not hasattr(lib, name):
        return None
    a = _as_u16(values)
    out = np.zeros(16, dtype=np.uint32)
    fn = getattr(lib, name)
    fn.argtypes = [_U16P, ctypes.c_size_t, _U32P]
    fn.restype = ctypes.c_int
    fn(_ptr16(a), a.size, out.ctypes.data_as(_U32P))
    return out


def generate(kind: int, seed: int, mask: int, first_index: int, n: int) -> np.ndarray:
    """Host twin of the product's on-device input makers."""
    out = np.empty(n, dtype=np.uint16)
    load_c().oracle_generate_u16(kind, seed, mask, first_index, n, _ptr16(out))
    return out


def flagstat_generated(kind: int, seed: int, mask: int, first_index: int, n: int,
                       threads: int | None = None) -> np.ndarray:
    """Counters of generate(kind, seed, mask, first_index, n) without holding it."""
    threads = threads or (os.cpu_count() or 1)
    out = np.zeros(32, dtype=np.uint64)
    load_c().oracle_flagstat_generated(kind, seed, mask, first_index, n, threads,
                                       out.ctypes.data_as(_U64P))
    return out


# --------------------------------------------------------------------------- #
# the reference itself (optional: oracle/_ref)
# --------------------------------------------------------------------------- #
def load_ref():
    """ctypes handle to oracle/_ref/libflagstats_ref.so, or None when absent."""
    global _ref
    if _ref is not None:
        return _ref
    path = os.path.join(_HERE, "_ref", "libflagstats_ref.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    for name in ("ref_FLAGSTAT_scalar", "ref_FLAGSTAT_sse4", "ref_FLAGSTAT_avx2",
                 "ref_FLAGSTAT_avx512", "ref_FLAGSTAT_avx512_improved3"):
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = [_U16P, ctypes.c_uint32, _U32P]
            fn.restype = ctypes.c_int
    lib.ref_FLAGSTATS_u16.argtypes = [_U16P, ctypes.c_uint32, _U32P]
    lib.ref_FLAGSTATS_u16.restype = ctypes.c_uint64
    lib.ref_dispatch_name.argtypes = [ctypes.c_uint32]
    lib.ref_dispatch_name.restype = ctypes.c_char_p
    for name in ("ref_dispatch_x64", "ref_scalar_x64"):
        fn = getattr(lib, name)
        fn.argtypes = [_U16P, ctypes.c_uint64, _U64P]
        fn.restype = None
    lib.ref_dispatch_repeat.argtypes = [_U16P, ctypes.c_uint64, ctypes.c_uint32, _U64P]
    lib.ref_dispatch_repeat.restype = None
    for name in ("ref_has_avx512bw", "ref_has_avx2", "ref_has_sse42", "ref_cpuid"):
        getattr(lib, name).restype = ctypes.c_int
    _ref = lib
    return lib


def ref_call(name: str, values, flags: np.ndarray | None = None) -> np.ndarray | None:
    """Call one of the reference's uint32-ABI kernels (e.g. 'FLAGSTAT_scalar').

    Accumulates into ``flags`` (uint32[32]) like the reference; returns None if
    the reference build or the ISA is unavailable on this host."""
    lib = load_ref()
    if lib is None or not hasattr(lib, "ref_" + name):
        return None
    a = _as_u16(values)
    if flags is None:
        flags = np.zeros(32, dtype=np.uint32)
    rc = getattr(lib, "ref_" + name)(_ptr16(a), ctypes.c_uint32(a.size),
                                     flags.ctypes.data_as(_U32P))
    if name != "FLAGSTATS_u16" and rc != 0:
        return None
    return flags
