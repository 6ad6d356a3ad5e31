"""``pyflagstats.flagstats`` -- Python entry point of the reference, on the MI355X engine.

Mirror of ``python/libflagstats.pyx:8-37``: same argument checks and messages,
same dict layout, same derived fields, same value type (``numpy.uint32``); the
32 counters come from ``libflagstats_hip.so`` (the C-ABI replacement of
``libflagstats.h:3024-3070``) instead of the header-only CPU kernels -- through
``FLAGSTATS_u16_x64``, the int-returning twin of ``FLAGSTATS_u16``, so that a GPU
failure raises here instead of aborting the interpreter (``FLAGSTATS_u16`` itself
has no error channel and aborts by default, as its reference callers ignore it).

Counter contents follow ``FLAGSTAT_scalar`` (``libflagstats.h:118-142``): only
the 19 live slots are ever non-zero.  On an x86 host the reference's dispatcher
additionally fills FPAIRED / FPROPER_PAIR / FMUNMAP / FREVERSE / FMREVERSE and a
pass-QC FQCFAIL count for the SIMD-covered prefix when n >= 256; those values are
ISA- and length-dependent and are not part of the contract.
"""
from __future__ import annotations

import numpy as np

from . import _lib

# python/libflagstats.pyx:24
SAM_FLAG_NAMES = ["FPAIRED", "FPROPER_PAIR", "FUNMAP", "FMUNMAP", "FREVERSE", "FMREVERSE", "FREAD1", "FREAD2",
                  "FSECONDARY", "FQCFAIL", "FDUP", "FSUPPLEMENTARY", "n_pair_good", "n_sgltn", "n_pair_map"]


def counters_u32(values: np.ndarray) -> np.ndarray:
    """32 ``uint32`` counters of a host ``uint16`` array, as ``FLAGSTATS_u16`` gives them (uint32 length, counters
    modulo 2^32)."""
    n = len(values)
    if n >= 2 ** 32:
        raise ValueError("FLAGSTATS_u16 takes a uint32 length; use flagstats_x64 for >= 2^32 flags")
    return counters_u64(values).astype("uint32")


def counters_u64(values: np.ndarray) -> np.ndarray:
    """32 ``uint64`` counters through the 64-bit entry point ``FLAGSTATS_u16_x64``."""
    out = np.zeros(32, dtype="uint64")
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_u16_x64(values.ctypes.data, len(values), out.ctypes.data), "FLAGSTATS_u16_x64")
    return out


def _validate(values):
    # python/libflagstats.pyx:9-17
    if type(values) != np.ndarray:  # noqa: E721  (the reference compares types exactly)
        raise ValueError("Values must be an numpy.ndarray")
    if values.dtype != "uint16":
        raise ValueError("Values must have the dtype \"uint16\"")
    if not values.flags['C_CONTIGUOUS']:
        print("Input array is not contiguous. Fixing...")
        values = np.ascontiguousarray(values, dtype=np.uint16)
    # the reference binds `uint16_t[::1] v = values` (pyx:21) and takes &v[0] (pyx:22)
    if values.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % values.ndim)
    if len(values) == 0:
        raise IndexError("Out of bounds on buffer access (axis 0)")
    return values


def _as_dict(flags: np.ndarray, n_values: int) -> dict:
    # python/libflagstats.pyx:26-35
    ret = {
        "n_values": n_values,
        "passed": dict(zip(SAM_FLAG_NAMES, flags[0:15, ])),
        "failed": dict(zip(SAM_FLAG_NAMES, flags[16:31, ])),
    }
    ret["passed"]["mapped"] = n_values - ret["passed"]["FUNMAP"] - ret["failed"]["FUNMAP"]
    ret["passed"]["paired_in_seq"] = ret["passed"]["FREAD1"] + ret["passed"]["FREAD2"]
    return ret


def flagstats(values):
    """Drop-in for ``pyflagstats.flagstats(values)`` (python/libflagstats.pyx:8)."""
    values = _validate(values)
    return _as_dict(counters_u32(values), len(values))


def flagstats_x64(values):
    """Same dict with ``numpy.uint64`` values; no 2^32 limit on length or counters."""
    values = _validate(values)
    return _as_dict(counters_u64(values), len(values))
