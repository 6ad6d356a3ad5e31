"""ctypes binding of the C-ABI library ``libflagstats_hip.so`` (include/libflagstats_hip.h; measurement entries:
include/libflagstats_hip_probe.h).

The library is built in-tree by ``__graft_entry__.build()`` (or
``make -C libflagstats_amd/csrc``).  If it is missing this module raises -- there
is no Python or CPU fallback for the hot path.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLAGSTATS_HIP_LIB: load another build of the same library (e.g. `make tuning` -> libflagstats_hip_tuning.so,
# the measurement build that also carries the K1 schedules that lost the sweeps)
LIB_PATH = os.environ.get("FLAGSTATS_HIP_LIB") or os.path.join(_HERE, "libflagstats_hip.so")

_U16P = ctypes.POINTER(ctypes.c_uint16)
_U32P = ctypes.POINTER(ctypes.c_uint32)
_U64P = ctypes.POINTER(ctypes.c_uint64)
FLAGSTATS_func = ctypes.CFUNCTYPE(ctypes.c_int, _U16P, ctypes.c_uint32, _U32P)

class BlockfileStats(ctypes.Structure):
    """FLAGSTATS_blockfile_stats of include/libflagstats_hip.h."""
    _fields_ = [("n_flags", ctypes.c_uint64), ("n_blocks", ctypes.c_uint64), ("compressed_bytes", ctypes.c_uint64),
                ("uncompressed_bytes", ctypes.c_uint64), ("wall_s", ctypes.c_double), ("index_s", ctypes.c_double),
                ("setup_s", ctypes.c_double), ("decode_cpu_s", ctypes.c_double),
                ("wait_decode_s", ctypes.c_double), ("wait_copy_s", ctypes.c_double), ("threads", ctypes.c_int32), ("chunks", ctypes.c_int32),
                ("gpu_decode", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class GpuLz4Stats(ctypes.Structure):
    """FLAGSTATS_gpu_lz4_stats of include/libflagstats_hip_probe.h."""
    _fields_ = [("n_blocks", ctypes.c_uint64), ("n_flags", ctypes.c_uint64), ("bad_blocks", ctypes.c_uint64),
                ("compressed_bytes", ctypes.c_uint64), ("decoded_bytes", ctypes.c_uint64),
                ("h2d_ms", ctypes.c_double), ("decode_ms", ctypes.c_double), ("count_ms", ctypes.c_double),
                ("sequences", ctypes.c_uint64), ("far_matches", ctypes.c_uint64), ("ring_kib", ctypes.c_uint64),
                ("chunks", ctypes.c_uint64), ("pipeline_ms", ctypes.c_double),
                ("uncompressed_bytes", ctypes.c_uint64), ("readers", ctypes.c_uint64), ("wall_s", ctypes.c_double),
                ("segments", ctypes.c_uint64)]


# name -> (restype, argtypes); mirrors include/libflagstats_hip.h + libflagstats_hip_probe.h one to one
SIGNATURES = {
    "FLAGSTATS_u16": (ctypes.c_uint64, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "FLAGSTATS_get_function": (FLAGSTATS_func, [ctypes.c_uint32]),
    "FLAGSTAT_hip": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "FLAGSTATS_u16_x64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16_store": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16_sync": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_u16_x64_superset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16_superset": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16_superset_sync": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_available": (ctypes.c_int, []),
    "FLAGSTATS_hip_device_count": (ctypes.c_int, []),
    "FLAGSTATS_hip_ctx_create": (ctypes.c_void_p, [ctypes.c_int]),
    "FLAGSTATS_hip_ctx_destroy": (None, [ctypes.c_void_p]),
    "FLAGSTATS_hip_ctx_device": (ctypes.c_int, [ctypes.c_void_p]),
    "FLAGSTATS_hip_ctx_u16_x64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_ctx_device_u16_sync": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_shard_range": (None, [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _U64P, _U64P]),
    "FLAGSTATS_hip_multi_u16_x64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                   ctypes.c_void_p]),
    "FLAGSTATS_hip_multi_device_u16": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _U64P, ctypes.c_int, ctypes.c_void_p]),
    "FLAGSTATS_hip_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "FLAGSTATS_hip_comm_init_rank": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "FLAGSTATS_hip_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "FLAGSTATS_hip_comm_count": (ctypes.c_int, [ctypes.c_void_p]),
    "FLAGSTATS_hip_comm_library": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]),
    "FLAGSTATS_hip_allreduce_counters": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_u16_allreduce_overlapped": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                                                     ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_stream_wait_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "FLAGSTATS_hip_device_u16_allreduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                                          ctypes.c_void_p]),
    "FLAGSTATS_hip_device_alloc_on": (ctypes.c_void_p, [ctypes.c_int, ctypes.c_size_t]),
    "FLAGSTATS_hip_init": (ctypes.c_int, [ctypes.c_int]),
    "FLAGSTATS_hip_shutdown": (None, []),
    "FLAGSTATS_hip_last_error": (ctypes.c_char_p, []),
    "FLAGSTATS_hip_device_id": (ctypes.c_int, []),
    "FLAGSTATS_hip_compute_units": (ctypes.c_int, []),
    "FLAGSTATS_hip_forked": (ctypes.c_int, []),
    "FLAGSTATS_hip_set": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_get": (ctypes.c_uint64, [ctypes.c_char_p]),
    "FLAGSTATS_text_to_u16": (ctypes.c_int64, [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]),
    "FLAGSTATS_text_count_lines": (ctypes.c_uint64, [ctypes.c_char_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_host_alloc": (ctypes.c_void_p, [ctypes.c_size_t]),
    "FLAGSTATS_hip_host_free": (None, [ctypes.c_void_p]),
    "FLAGSTATS_hip_device_alloc": (ctypes.c_void_p, [ctypes.c_size_t]),
    "FLAGSTATS_hip_device_free": (None, [ctypes.c_void_p]),
    "FLAGSTATS_hip_memcpy_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "FLAGSTATS_hip_memcpy_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "FLAGSTATS_hip_synchronize": (ctypes.c_int, []),
    "FLAGSTATS_hip_generate_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64,
                                                  ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_time_device_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                     ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]),
    "FLAGSTATS_hip_time_device_u16_rotating": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                                              ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                                              ctypes.c_void_p]),
    "FLAGSTATS_hip_sclk_under_load": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "FLAGSTATS_hip_blockfile_lz4": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p,
                                                   ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_blockimage_lz4": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                                    ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_blockimage_lz4_gpu": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(GpuLz4Stats)]),
    "FLAGSTATS_hip_blockimage_zstd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                                     ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_blockfile_zstd": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_blockfile": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_zstd_available": (ctypes.c_int, []),
    "FLAGSTATS_hip_blockfile_superset": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_file_raw_superset": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_file_raw": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_hip_host_staged_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(BlockfileStats)]),
    "FLAGSTATS_lz4_block_decode": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_stream_open": (ctypes.c_void_p, []),
    "FLAGSTATS_hip_stream_acquire": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_stream_commit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_stream_push": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]),
    "FLAGSTATS_hip_stream_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "FLAGSTATS_hip_stream_flags": (ctypes.c_uint64, [ctypes.c_void_p]),
    "FLAGSTATS_hip_stream_close": (None, [ctypes.c_void_p]),
    "STORM_pospopcnt_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "FLAGSTATS_hip_pospopcnt_u16_x64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "FLAGSTATS_hip_device_pospopcnt_u16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                                          ctypes.c_void_p]),
    "FLAGSTATS_hip_read_probe": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.POINTER(ctypes.c_float)]),
    "FLAGSTATS_hip_read_probe_policy": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.POINTER(ctypes.c_float)]),
    "FLAGSTATS_hip_read_probe2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


class FlagstatsHipError(RuntimeError):
    """The GPU path failed; there is no CPU fallback to hide it."""


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64; a process must not end up with torch
    bound to a different HIP runtime than the one that is already loaded.  If torch is installed
    but not imported yet, load ITS runtime first (by SONAME both then resolve to the same copy);
    importing torch later then works in either order.  No torch: the system runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("FLAGSTATS_HIP_SYSTEM_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> ctypes.CDLL:
    """Load libflagstats_hip.so (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        if _lib.FLAGSTATS_hip_forked():
            # the C entry points refuse a forked child themselves; raising here names the remedy before any call is made
            raise FlagstatsHipError(
                "this process (pid %d) was fork()ed after libflagstats_hip had been used: a HIP context, its streams and the "
                "library's worker threads do not exist in a forked child. Start workers with the \"spawn\" start method "
                "(multiprocessing.get_context(\"spawn\")), or make the first call after the fork." % os.getpid())
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C libflagstats_amd/csrc`). "
            "libflagstats_amd has no CPU fallback for the flagstat hot path.")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here = header/library drift
        fn.restype = restype
        fn.argtypes = argtypes
    # The library's process-wide "on_error" policy is left alone: the reference-shaped entry points (FLAGSTATS_u16,
    # FLAGSTAT_hip, STORM_pospopcnt_u16) abort() after a failure by default because their reference callers ignore the
    # return value -- another consumer of the same .so in this process (the reference's own .pyx, say) must keep that
    # protection.  This package calls the int-returning 64-bit entry points instead and raises FlagstatsHipError.
    _lib = handle
    return handle


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().FLAGSTATS_hip_last_error().decode(errors="replace")
        raise FlagstatsHipError(f"{what} failed (rc={rc}): {msg}")
