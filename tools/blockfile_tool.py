#!/usr/bin/env python3
"""Writer for the reference's block-file format + liblz4 oracle bindings (tests / benches only).

The reference's ``bench compress`` (benchmark/flagstats.cpp:110-190) cuts the raw uint16 stream into
1,024,000-byte blocks and writes ``int32 uncompressed_size, int32 compressed_size, <LZ4 block>`` per
block with liblz4's LZ4_compress_fast (``lz4f``) or LZ4_compress_HC (``lz4hc``); when the input
size is a multiple of the block size its read loop emits one more, empty, block.  This tool does
the same through the image's own liblz4.so.1 (ctypes), so the files the product's reader is tested
on come from the real compressor, not from our code.
"""
import ctypes
import ctypes.util
import struct
import sys

import numpy as np

BLOCK_BYTES = 1024000  # benchmark/flagstats.cpp:119

_lz4 = None


def lz4():
    global _lz4
    if _lz4 is None:
        name = ctypes.util.find_library("lz4") or "liblz4.so.1"
        lib = ctypes.CDLL(name)
        lib.LZ4_compressBound.restype = ctypes.c_int
        lib.LZ4_compressBound.argtypes = [ctypes.c_int]
        lib.LZ4_compress_fast.restype = ctypes.c_int
        lib.LZ4_compress_fast.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.LZ4_compress_HC.restype = ctypes.c_int
        lib.LZ4_compress_HC.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.LZ4_decompress_safe.restype = ctypes.c_int
        lib.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.LZ4_versionNumber.restype = ctypes.c_int
        _lz4 = lib
    return _lz4


_zstd = None


def zstd():
    """The image's libzstd.so.1 (what benchmark/flagstats.cpp:85-93 calls): ZSTD_compress / ZSTD_decompress."""
    global _zstd
    if _zstd is None:
        lib = ctypes.CDLL(ctypes.util.find_library("zstd") or "libzstd.so.1")
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_isError.restype = ctypes.c_uint
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _zstd = lib
    return _zstd


def compress_block(raw: bytes, mode: str = "fast", level: int = 2) -> bytes:
    if mode == "zstd":
        z = zstd()
        bound = z.ZSTD_compressBound(len(raw))
        dst = ctypes.create_string_buffer(max(bound, 1))
        n = z.ZSTD_compress(dst, bound, raw, len(raw), level)             # benchmark/flagstats.cpp:86
        if z.ZSTD_isError(n):
            raise RuntimeError("libzstd compression failed")
        return dst.raw[:n]
    lib = lz4()
    bound = lib.LZ4_compressBound(len(raw))
    dst = ctypes.create_string_buffer(max(bound, 1))
    if mode == "fast":
        n = lib.LZ4_compress_fast(raw, dst, len(raw), bound, level)      # benchmark/flagstats.cpp:127
    else:
        n = lib.LZ4_compress_HC(raw, dst, len(raw), bound, level)        # benchmark/flagstats.cpp:167
    if n <= 0:
        raise RuntimeError("liblz4 compression failed")
    return dst.raw[:n]


def decompress_block_ref(comp: bytes, usize: int):
    """liblz4's LZ4_decompress_safe (what benchmark/flagstats.cpp:316 calls): bytes or None."""
    dst = ctypes.create_string_buffer(max(usize, 1))
    n = lz4().LZ4_decompress_safe(comp, dst, len(comp), usize)
    return None if n < 0 else dst.raw[:n]


def block_file_image(flags: np.ndarray, block_bytes: int = BLOCK_BYTES, mode: str = "fast", level: int = 2,
                     trailing_empty: bool = True) -> bytes:
    raw = np.ascontiguousarray(flags, dtype=np.uint16).tobytes()
    out = []
    pos = 0
    while pos < len(raw):
        chunk = raw[pos:pos + block_bytes]
        comp = compress_block(chunk, mode, level)
        out.append(struct.pack("<ii", len(chunk), len(comp)))
        out.append(comp)
        pos += len(chunk)
    if trailing_empty and len(raw) % block_bytes == 0:
        comp = compress_block(b"", mode, level)   # the reference's loop runs once more on a 0-byte read
        out.append(struct.pack("<ii", 0, len(comp)))
        out.append(comp)
    return b"".join(out)


def write_block_file(path, flags, **kw) -> int:
    img = block_file_image(flags, **kw)
    with open(path, "wb") as f:
        f.write(img)
    return len(img)


if __name__ == "__main__":
    # usage: blockfile_tool.py raw_u16.bin out.lz4 [fast|hc] [level]
    a = np.fromfile(sys.argv[1], dtype=np.uint16)
    n = write_block_file(sys.argv[2], a, mode=sys.argv[3] if len(sys.argv) > 3 else "fast",
                         level=int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    print("%d flags -> %d bytes" % (a.size, n))
