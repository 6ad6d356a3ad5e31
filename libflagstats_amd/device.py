"""Device-resident FLAG arrays: allocation, on-device generation and counting.

Everything here goes through the C-ABI of ``libflagstats_hip.so``; torch is
optional plumbing (device memory that the caller already owns, its current
stream, ``torch.distributed``), never the thing that computes.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

GEN_UNIFORM, GEN_NA12878, GEN_RAMP = 0, 1, 2


class DeviceFlags:
    """A ``uint16`` FLAG array in HBM owned by the library (``hipMalloc``)."""

    def __init__(self, n: int):
        self.n = int(n)
        self._lib = _lib.lib()
        self.ptr = self._lib.FLAGSTATS_hip_device_alloc(max(self.n, 1) * 2)
        if not self.ptr:
            _lib.check(-1, "FLAGSTATS_hip_device_alloc(%d bytes)" % (self.n * 2))

    def free(self) -> None:
        if self.ptr:
            self._lib.FLAGSTATS_hip_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # -- data movement -----------------------------------------------------
    def upload(self, values: np.ndarray, offset: int = 0) -> "DeviceFlags":
        v = np.ascontiguousarray(values, dtype=np.uint16)
        assert offset + v.size <= self.n
        _lib.check(self._lib.FLAGSTATS_hip_memcpy_h2d(self.ptr + 2 * offset, v.ctypes.data, v.size * 2), "memcpy_h2d")
        return self

    def download(self, offset: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n - offset if n is None else n
        out = np.empty(n, dtype=np.uint16)
        if n:
            _lib.check(self._lib.FLAGSTATS_hip_memcpy_d2h(out.ctypes.data, self.ptr + 2 * offset, n * 2), "memcpy_d2h")
        return out

    def generate(self, kind: int, seed: int, mask: int = 0xFFFF, first_index: int = 0, offset: int = 0,
                 n: int | None = None, stream: int | None = None) -> "DeviceFlags":
        """Fill [offset, offset+n) with flag(kind, seed, mask, first_index + k) on device."""
        n = self.n - offset if n is None else n
        _lib.check(self._lib.FLAGSTATS_hip_generate_u16(self.ptr + 2 * offset, n, kind, seed, mask, first_index,
                                                        stream), "FLAGSTATS_hip_generate_u16")
        if stream is None:
            _lib.check(self._lib.FLAGSTATS_hip_synchronize(), "synchronize")
        return self

    # -- counting ----------------------------------------------------------
    def count(self, offset: int = 0, n: int | None = None) -> np.ndarray:
        """uint64[32] counters of [offset, offset+n): one K1 launch on device (atomic epilogue), synchronous."""
        n = self.n - offset if n is None else n
        return count_device_ptr(self.ptr + 2 * offset, n)


def pospopcnt_host(values: np.ndarray) -> np.ndarray:
    """``STORM_pospopcnt_u16`` (python/libalgebra.h:3496-3551) of a host array: uint32[16], zeroed first."""
    v = np.ascontiguousarray(values, dtype=np.uint16)
    wide = np.zeros(16, dtype=np.uint64)             # the int-returning twin: a failure raises, never aborts
    _lib.check(_lib.lib().FLAGSTATS_hip_pospopcnt_u16_x64(v.ctypes.data if v.size else None, v.size, wide.ctypes.data),
               "FLAGSTATS_hip_pospopcnt_u16_x64")
    return wide.astype(np.uint32)


def pospopcnt_torch(t, out=None):
    """int64[16] += positional popcount of a 16-bit CUDA tensor, on torch's current stream."""
    import torch

    assert t.is_cuda and t.is_contiguous() and t.element_size() == 2
    if out is None:
        out = torch.zeros(16, dtype=torch.int64, device=t.device)
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _lib.check(_lib.lib().FLAGSTATS_hip_device_pospopcnt_u16(t.data_ptr(), t.numel(), out.data_ptr(),
                                                             ctypes.c_void_p(stream)), "FLAGSTATS_hip_device_pospopcnt_u16")
    return out


def count_device_ptr(ptr: int, n: int) -> np.ndarray:
    """uint64[32] counters of a device ``uint16`` array given as a raw pointer."""
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(_lib.lib().FLAGSTATS_hip_device_u16_sync(ptr, n, out.ctypes.data), "FLAGSTATS_hip_device_u16_sync")
    return out


def count_device_async(ptr: int, n: int, d_out_ptr: int, stream: int | None) -> None:
    """d_out[32] (device uint64) += counters, asynchronously on ``stream``."""
    _lib.check(_lib.lib().FLAGSTATS_hip_device_u16(ptr, n, d_out_ptr, stream), "FLAGSTATS_hip_device_u16")


def count_torch(t, out=None, store: bool = False):
    """Counters of a CUDA torch tensor of 16-bit elements, on torch's current stream.

    Returns (or adds into) an ``int64[32]`` CUDA tensor; nothing is synchronised.
    ``store=True`` overwrites ``out`` instead of accumulating (no zeroing launch needed).
    """
    import torch

    assert t.is_cuda and t.is_contiguous() and t.element_size() == 2, "need a contiguous 16-bit CUDA tensor"
    if out is None:
        out = torch.zeros(32, dtype=torch.int64, device=t.device)
    assert out.is_cuda and out.dtype == torch.int64 and out.numel() == 32 and out.is_contiguous()
    stream = ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    if store:
        _lib.check(_lib.lib().FLAGSTATS_hip_device_u16_store(t.data_ptr(), t.numel(), out.data_ptr(), stream),
                   "FLAGSTATS_hip_device_u16_store")
    else:
        count_device_async(t.data_ptr(), t.numel(), out.data_ptr(), stream)
    return out


def generate_torch(t, kind: int, seed: int, mask: int = 0xFFFF, first_index: int = 0):
    """Fill a 16-bit CUDA torch tensor on torch's current stream."""
    import torch

    assert t.is_cuda and t.is_contiguous() and t.element_size() == 2
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _lib.check(_lib.lib().FLAGSTATS_hip_generate_u16(t.data_ptr(), t.numel(), kind, seed, mask, first_index,
                                                     ctypes.c_void_p(stream)), "FLAGSTATS_hip_generate_u16")
    return t


def time_device_ptr(ptr: int, n: int, warmup: int, reps: int):
    """(ms_total, counters-of-one-pass) for `reps` back-to-back launches of the hot path between hipEvents."""
    ms = ctypes.c_float(0.0)
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(_lib.lib().FLAGSTATS_hip_time_device_u16(ptr, n, warmup, reps, ctypes.byref(ms), out.ctypes.data),
               "FLAGSTATS_hip_time_device_u16")
    return float(ms.value), out


def time_device_rotating(ptr: int, n: int, stride_flags: int, slots: int, warmup: int, reps: int):
    """(ms_total, counters summed over the timed launches): back-to-back launches, launch i on slice (i * 7919) % slots."""
    ms = ctypes.c_float(0.0)
    out = np.zeros(32, dtype=np.uint64)
    _lib.check(_lib.lib().FLAGSTATS_hip_time_device_u16_rotating(ptr, n, stride_flags, slots, warmup, reps, ctypes.byref(ms),
                                                               out.ctypes.data), "FLAGSTATS_hip_time_device_u16_rotating")
    return float(ms.value), out
