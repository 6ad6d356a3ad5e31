"""Block files of the reference's bench programs, counted on the MI355X engine (row f1).

``flagstat_lz4_file`` is the counterpart of ``bench decompress -i X.lz4 -d``
(``benchmark/flagstats.cpp:288-358``): threaded host LZ4 block decode overlapped with the H2D
copy and K1/K2.  ``flagstat_raw_file`` is ``-D`` (``:415-468``).  Both return
``(uint64[32] counters, stats dict)``.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


def _stats(st: _lib.BlockfileStats) -> dict:
    return {name: getattr(st, name) for name, _ in st._fields_}


def _image_pointer(image):
    """(pointer argument, byte count, object to keep alive) for a file image WITHOUT copying it: `bytes` go to the C side as they
    are, numpy arrays by their data pointer, other buffers (bytearray, memoryview, mmap) through the buffer protocol; only a
    read-only buffer that is neither is copied (a 30 MB image: 3 ms, more than the call takes)."""
    if image is None or len(image) == 0:
        return None, 0, None
    if isinstance(image, bytes):
        return image, len(image), image
    if isinstance(image, np.ndarray):
        a = np.ascontiguousarray(image)
        return a.ctypes.data, a.nbytes, a
    mv = memoryview(image)
    if mv.readonly or not mv.contiguous:
        b = mv.tobytes()
        return b, len(b), b
    buf = (ctypes.c_char * mv.nbytes).from_buffer(mv)
    return buf, mv.nbytes, buf


def flagstat_lz4_file(path: str, threads: int = 0):
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    _lib.check(_lib.lib().FLAGSTATS_hip_blockfile_lz4(str(path).encode(), threads, out.ctypes.data, ctypes.byref(st)),
               "FLAGSTATS_hip_blockfile_lz4")
    return out, _stats(st)


def flagstat_lz4_image(image: bytes, threads: int = 0):
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    buf, nbytes, _keep = _image_pointer(image)
    _lib.check(_lib.lib().FLAGSTATS_hip_blockimage_lz4(buf, nbytes, threads, out.ctypes.data, ctypes.byref(st)),
               "FLAGSTATS_hip_blockimage_lz4")
    return out, _stats(st)


def flagstat_zstd_file(path: str, threads: int = 0):
    """``.zst`` block file (``benchmark/flagstats.cpp:636-682``).  Files of 64 MiB and more are decoded on the GPU (knob
    ``zstd_decoder``; ``stats["gpu_decode"]`` says which ran); smaller ones, and files with frames the GPU decoder does not
    take, by libzstd.so.1 on host threads (resolved at run time)."""
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    _lib.check(_lib.lib().FLAGSTATS_hip_blockfile_zstd(str(path).encode(), threads, out.ctypes.data, ctypes.byref(st)),
               "FLAGSTATS_hip_blockfile_zstd")
    return out, _stats(st)


def flagstat_zstd_image(image: bytes, threads: int = 0):
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    buf, nbytes, _keep = _image_pointer(image)
    _lib.check(_lib.lib().FLAGSTATS_hip_blockimage_zstd(buf, nbytes, threads, out.ctypes.data, ctypes.byref(st)),
               "FLAGSTATS_hip_blockimage_zstd")
    return out, _stats(st)


def flagstat_file(path: str, threads: int = 0, superset: bool = False):
    """Codec by extension (``.lz4`` / ``.zst``), as the reference's ``check_file_extension`` (``:828-839``).
    ``superset=True``: also slots 0 / 16 (n_pair_all) and 9 (pass-QC reads), for the samtools report."""
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    fn = _lib.lib().FLAGSTATS_hip_blockfile_superset if superset else _lib.lib().FLAGSTATS_hip_blockfile
    _lib.check(fn(str(path).encode(), threads, out.ctypes.data, ctypes.byref(st)), "FLAGSTATS_hip_blockfile")
    return out, _stats(st)


def flagstat_raw_file(path: str, superset: bool = False):
    out = np.zeros(32, dtype=np.uint64)
    st = _lib.BlockfileStats()
    fn = _lib.lib().FLAGSTATS_hip_file_raw_superset if superset else _lib.lib().FLAGSTATS_hip_file_raw
    _lib.check(fn(str(path).encode(), out.ctypes.data, ctypes.byref(st)), "FLAGSTATS_hip_file_raw")
    return out, _stats(st)


def lz4_block_decode(src: bytes, dstcap: int):
    """The product's host LZ4 block decoder; returns bytes or None on malformed input."""
    dst = ctypes.create_string_buffer(max(dstcap, 1))
    n = _lib.lib().FLAGSTATS_lz4_block_decode(src, len(src), dst, dstcap)
    return None if n < 0 else dst.raw[:n]
