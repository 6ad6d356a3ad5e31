"""Streaming sessions (``FLAGSTATS_hip_stream_*``): counting behind a caller-owned per-block loop.

The shape of the reference's block reader (``benchmark/flagstats.cpp:311-342``): one counter
array accumulated over all blocks, read after the loop.  ``acquire(n)`` hands out pinned memory
to decode into (a numpy view, zero copy), ``commit(n)`` queues it, ``finish()`` returns the
counters of everything committed since the last finish.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


class StreamSession:
    def __init__(self):
        self._lib = _lib.lib()
        self._h = self._lib.FLAGSTATS_hip_stream_open()
        if not self._h:
            _lib.check(-1, "FLAGSTATS_hip_stream_open")

    def acquire(self, n: int) -> np.ndarray:
        """uint16[n] view of pinned memory, valid until commit()."""
        p = self._lib.FLAGSTATS_hip_stream_acquire(self._h, n)
        if not p:
            _lib.check(-1, "FLAGSTATS_hip_stream_acquire")
        buf = (ctypes.c_uint16 * n).from_address(p)
        return np.frombuffer(buf, dtype=np.uint16, count=n)

    def acquire_ptr(self, n: int) -> int:
        p = self._lib.FLAGSTATS_hip_stream_acquire(self._h, n)
        if not p:
            _lib.check(-1, "FLAGSTATS_hip_stream_acquire")
        return p

    def commit(self, n: int) -> None:
        _lib.check(self._lib.FLAGSTATS_hip_stream_commit(self._h, n), "FLAGSTATS_hip_stream_commit")

    def push(self, values: np.ndarray) -> None:
        v = np.ascontiguousarray(values, dtype=np.uint16)
        _lib.check(self._lib.FLAGSTATS_hip_stream_push(self._h, v.ctypes.data if v.size else None, v.size),
                   "FLAGSTATS_hip_stream_push")

    @property
    def pending_flags(self) -> int:
        return int(self._lib.FLAGSTATS_hip_stream_flags(self._h))

    def finish(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros(32, dtype=np.uint64)
        _lib.check(self._lib.FLAGSTATS_hip_stream_finish(self._h, out.ctypes.data), "FLAGSTATS_hip_stream_finish")
        return out

    def close(self) -> None:
        if self._h:
            self._lib.FLAGSTATS_hip_stream_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
