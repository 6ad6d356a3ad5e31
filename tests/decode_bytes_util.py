"""The two GPU block decoders called DIRECTLY (fsk_lz4_decode / fsk_zstd_decode, the launch entries the block-file
orchestration itself calls) so that a test can look at the BYTES they write, not at counters derived from them:
device buffers through the library's own C-ABI (no torch), reference decodes through the image's liblz4 / libzstd --
what the reference calls (benchmark/flagstats.cpp:316 LZ4_decompress_safe, :96 ZSTD_decompress).
Test infrastructure: tests/test_gpu_decode_bytes.py and the fuzzers under tests/perf/ use it."""
import ctypes
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import blockfile_tool as bt  # noqa: E402

FILL = 0xA5     # what the output buffer holds before a launch: a byte the decoder did not write still holds it
NOT_RUN = 0xFFFFFFFF


class GpuBlock(ctypes.Structure):   # fsk::GpuBlock, libflagstats_amd/csrc/flagstat_lz4_kernels.h
    _fields_ = [("src_off", ctypes.c_uint64), ("dst_off", ctypes.c_uint64), ("src_len", ctypes.c_uint32), ("dst_len", ctypes.c_uint32)]


def _protos(lib):
    if getattr(lib, "_fsk_protos", False):
        return
    lib.fsk_lz4_decode.restype = ctypes.c_int
    lib.fsk_lz4_decode.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.fsk_zstd_scratch_bytes.restype = ctypes.c_uint64
    lib.fsk_zstd_scratch_bytes.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    lib.fsk_zstd_decode.restype = ctypes.c_int
    lib.fsk_zstd_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    lib.fsk_zstd_scratch_bytes_ex.restype = ctypes.c_uint64
    lib.fsk_zstd_scratch_bytes_ex.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.fsk_zstd_decode_ex.restype = ctypes.c_int
    lib.fsk_zstd_decode_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    lib._fsk_protos = True


class DeviceDecode:
    """One launch of a decode kernel over `payloads` (compressed bytes) with declared decoded `sizes`; the decoded buffer
    stays on the device until free() so that further device entries (pospopcnt, K1) can run on what the decoder wrote.
    codec: "lz4" (kernel 0 = the workgroup pipeline, 1 = one wave per block) or "zstd" (min_blocks > 0: the second-pass entry
    fsk_zstd_decode_ex, with room for that many Zstandard blocks per frame)."""

    def __init__(self, lib, codec, payloads, sizes, kernel=0, min_blocks=0):
        from libflagstats_amd import _lib
        _protos(lib)
        self.lib, self._check = lib, _lib.check
        self.n = n = len(payloads)
        self.sizes = [int(s) for s in sizes]
        self._ptrs = []
        blocks = (GpuBlock * max(n, 1))()
        comp = bytearray(16)
        dpos = 0
        for i, (p, sz) in enumerate(zip(payloads, self.sizes)):
            comp += b"\0" * 8                          # (the block header's place in a file)
            blocks[i] = GpuBlock(len(comp), dpos, len(p), sz)
            comp += p
            dpos += (sz + 15) & ~15
        comp += b"\0" * 64                             # the kernels may read 64 bytes past the last payload
        self.blocks, self.out_bytes = blocks, dpos + 64
        self.d_comp = self._up(bytes(comp))
        self.d_blocks = self._up(bytes(blocks))
        self.d_out = self._up(bytes([FILL]) * self.out_bytes)
        self.d_status = self._up(b"\xff" * (4 * max(n, 1)))
        self.d_tally = self._up(bytes(8 * 32))
        if codec == "zstd":
            mx = max(self.sizes) if n else 0
            need = lib.fsk_zstd_scratch_bytes_ex(mx, n, min_blocks)
            d_scratch = self._alloc(need + 256)
            if min_blocks:
                rc = lib.fsk_zstd_decode_ex(self.d_comp, self.d_blocks, n, self.d_out, self.d_status, self.d_tally, (d_scratch + 255) & ~255, need, mx, min_blocks, 0, None)
            else:
                rc = lib.fsk_zstd_decode(self.d_comp, self.d_blocks, n, self.d_out, self.d_status, self.d_tally, (d_scratch + 255) & ~255, need, mx, 0, None)
        else:
            rc = lib.fsk_lz4_decode(kernel, self.d_comp, self.d_blocks, n, self.d_out, self.d_status, self.d_tally, 0, None)
        if rc:
            raise RuntimeError("fsk_%s_decode: hip error %d" % (codec, rc))
        self._check(lib.FLAGSTATS_hip_synchronize(), "synchronize")
        self.status = np.frombuffer(self._down(self.d_status, 4 * max(n, 1)), dtype=np.uint32)[:n].copy()
        self.tally = np.frombuffer(self._down(self.d_tally, 8 * 32), dtype=np.uint64).copy()
        self.out = self._down(self.d_out, self.out_bytes)

    def _alloc(self, nbytes):
        p = self.lib.FLAGSTATS_hip_device_alloc(max(int(nbytes), 16))
        if not p:
            self._check(-1, "FLAGSTATS_hip_device_alloc(%d)" % nbytes)
        self._ptrs.append(p)
        return p

    def _up(self, data):
        p = self._alloc(len(data))
        if data:
            buf = ctypes.create_string_buffer(data, len(data))
            self._check(self.lib.FLAGSTATS_hip_memcpy_h2d(p, buf, len(data)), "memcpy_h2d")
        return p

    def _down(self, p, nbytes):
        buf = ctypes.create_string_buffer(max(nbytes, 1))
        if nbytes:
            self._check(self.lib.FLAGSTATS_hip_memcpy_d2h(buf, p, nbytes), "memcpy_d2h")
        return buf.raw[:nbytes]

    def slot(self, i):
        """the whole 16-byte-padded slot of block i (decoded bytes, then whatever the buffer held before)"""
        b = self.blocks[i]
        end = self.blocks[i + 1].dst_off if i + 1 < self.n else self.out_bytes
        return self.out[b.dst_off:end]

    def pospopcnt(self, i):
        """FLAGSTATS_hip_device_pospopcnt_u16 (all 16 bit positions) over block i's flags where the decoder left them"""
        b = self.blocks[i]
        d_cnt = self._up(bytes(8 * 16))
        self._check(self.lib.FLAGSTATS_hip_device_pospopcnt_u16(self.d_out + b.dst_off, b.dst_len >> 1, d_cnt, None), "device_pospopcnt_u16")
        self._check(self.lib.FLAGSTATS_hip_synchronize(), "synchronize")
        return np.frombuffer(self._down(d_cnt, 8 * 16), dtype=np.uint64).copy()

    def free(self):
        for p in self._ptrs:
            self.lib.FLAGSTATS_hip_device_free(p)
        self._ptrs = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()


def pospopcnt_ref(raw):
    """the rule of STORM_pospopcnt_u16 (python/libalgebra.h:566-574): out[j] = words with bit j set"""
    a = np.frombuffer(raw[:len(raw) & ~1], dtype=np.uint16)
    return np.array([int(((a >> j) & 1).sum()) for j in range(16)], dtype=np.uint64)


def ref_lz4(comp, usize):
    """liblz4's LZ4_decompress_safe: bytes, or None when it rejects the block / decodes to another size"""
    r = bt.decompress_block_ref(bytes(comp), usize)
    return r if r is not None and len(r) == usize else None


def ref_zstd(comp, usize):
    """libzstd's ZSTD_decompress into exactly `usize` bytes: bytes or None"""
    z = bt.zstd()
    dst = ctypes.create_string_buffer(max(usize, 1))
    r = z.ZSTD_decompress(dst, usize, bytes(comp), len(comp))
    if z.ZSTD_isError(r) or r != usize:
        return None
    return dst.raw[:usize]


def split_image(img):
    """a block-file image (int32 usize, int32 csize, payload; benchmark/flagstats.cpp:119-138) -> [(payload, usize)]"""
    out, pos = [], 0
    while pos < len(img):
        us, cs = struct.unpack_from("<ii", img, pos)
        out.append((bytes(img[pos + 8:pos + 8 + cs]), us))
        pos += 8 + cs
    return out


def check_decoded(dd, i, want, what):
    """block i of a finished DeviceDecode against the reference decoder's bytes.  The decoders drop an odd trailing byte of
    a block (it is no flag: N = size >> 1, benchmark/flagstats.cpp:323) and must write NOTHING behind the even length."""
    assert dd.status[i] == 0, (what, i, "status", int(dd.status[i]))
    slot = dd.slot(i)
    even = len(want) & ~1
    if slot[:even] != want[:even]:
        g = np.frombuffer(slot[:even], dtype=np.uint8)
        w = np.frombuffer(want[:even], dtype=np.uint8)
        d = np.nonzero(g != w)[0]
        raise AssertionError("%s, block %d: %d of %d bytes differ, first at %d (got %s want %s)" % (
            what, i, len(d), even, d[0], g[d[0]:d[0] + 8].tolist(), w[d[0]:d[0] + 8].tolist()))
    rest = slot[even:]
    assert rest == bytes([FILL]) * len(rest), (what, i, "bytes written behind the block's even length")
