"""Frame generators of the Zstandard decoder's differential fuzz (tests/perf/fuzz_zstd_gpu.py) and of the byte-exact decode
tests (tests/test_gpu_decode_bytes.py): synthetic payloads of many shapes, ZSTD_compress2 with random advanced parameters
(frames ZSTD_compress would never write at any level), damage."""
import ctypes

import numpy as np


def synthetic(rng, nrng):
    import oracle
    kind = rng.randrange(9)
    n = rng.choice([0, 1, 2, 7, 100, 4097, 70000, 131072, 131073, 262144 + 5, 400000])
    if kind == 0:
        return oracle.generate(oracle.GEN_NA12878, rng.randrange(1000), 1, 0, (n + 1) // 2).tobytes()[:n]
    if kind == 1:
        return nrng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 2:
        return bytes([rng.randrange(256)]) * n
    if kind == 3:
        return nrng.integers(0, rng.choice([2, 4, 17, 60]), n, dtype=np.uint8).tobytes()
    if kind == 4:
        a = nrng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
        return (a + bytes(rng.randrange(1, 60000)) + a[:rng.randrange(1, 30000)] + nrng.integers(0, 256, rng.randrange(1, 20000), dtype=np.uint8).tobytes() + a)[:max(n, 50000)]
    if kind == 5:
        word = bytes(nrng.integers(0, 256, rng.randrange(1, 40), dtype=np.uint8))
        return (word * (n // len(word) + 1))[:n]
    if kind == 6:
        return nrng.integers(0, 3000, (n + 1) // 2, dtype=np.uint16).tobytes()[:n]
    if kind == 7:
        # far matches: a long random stretch repeated at a distance above the 64 KiB the execution kernel keeps in LDS
        a = nrng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
        return a + nrng.integers(0, 256, 90000, dtype=np.uint8).tobytes() + a + bytes(1000) + a[100:4000]
    return oracle.generate(oracle.GEN_UNIFORM, rng.randrange(1000), 0x0FFF, 0, (n + 1) // 2).tobytes()[:n]


def compress_with_parameters(z, rng, raw):
    """ZSTD_compress2 with random advanced parameters (strategy, window / hash / chain / search logs, minimum match, target
    length, long-distance matching): frames ZSTD_compress would never write at any level"""
    z.ZSTD_createCCtx.restype = ctypes.c_void_p
    z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
    z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    z.ZSTD_compress2.restype = ctypes.c_size_t
    cctx = z.ZSTD_createCCtx()
    # ZSTD_cParameter: compressionLevel 100, windowLog 101, hashLog 102, chainLog 103, searchLog 104, minMatch 105, targetLength 106,
    # strategy 107, enableLongDistanceMatching 160, ldmHashLog 161, ldmMinMatch 162, contentSizeFlag 200, checksumFlag 201
    picks = [(100, rng.choice([1, 3, 6, 12, 19])), (107, rng.randrange(1, 10)), (101, rng.randrange(10, 24)), (105, rng.randrange(3, 8)),
             (106, rng.choice([0, 4, 16, 64, 999])), (102, rng.randrange(6, 20)), (103, rng.randrange(6, 20)), (104, rng.randrange(1, 8)),
             (160, rng.randrange(2)), (200, rng.randrange(2)), (201, 0)]
    for key, value in picks:
        if rng.randrange(3):
            z.ZSTD_CCtx_setParameter(cctx, key, value)   # (out-of-range combinations are refused by the library: ignored here)
    bound = z.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(bound)
    n = z.ZSTD_compress2(cctx, dst, bound, raw, len(raw))
    z.ZSTD_freeCCtx(cctx)
    if z.ZSTD_isError(n):
        return None
    return dst.raw[:n]


def damage(rng, comp):
    bad = bytearray(comp)
    how = rng.randrange(4)
    if how == 0 or len(bad) < 20:
        for _ in range(rng.randrange(1, 4)):
            bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
    elif how == 1:
        del bad[rng.randrange(len(bad)):]
    elif how == 2:
        i = rng.randrange(len(bad))
        bad[i:i + rng.randrange(1, 8)] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
    else:
        i, j = sorted((rng.randrange(len(bad)), rng.randrange(len(bad))))
        bad[i:j] = bad[i:j][::-1]
    return bytes(bad) if bad else b"\0"


def flushed_frame(z, raw, every):
    """one frame from a streaming compressor that flushes every `every` bytes: a Zstandard block per flush, more and smaller
    blocks than ZSTD_compress makes (z = blockfile_tool.zstd())"""
    z.ZSTD_createCCtx.restype = ctypes.c_void_p
    z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    z.ZSTD_compressStream2.restype = ctypes.c_size_t

    class Buf(ctypes.Structure):
        _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]

    z.ZSTD_compressStream2.argtypes = [ctypes.c_void_p, ctypes.POINTER(Buf), ctypes.POINTER(Buf), ctypes.c_int]
    cctx = z.ZSTD_createCCtx()
    dst = ctypes.create_string_buffer(len(raw) + len(raw) // every * 32 + 1024)
    src = ctypes.create_string_buffer(raw, len(raw))
    ob = Buf(ctypes.cast(dst, ctypes.c_void_p), len(dst), 0)
    at = 0
    while at < len(raw):
        n = min(every, len(raw) - at)
        ib = Buf(ctypes.cast(src, ctypes.c_void_p).value + at, n, 0)
        last = at + n == len(raw)
        while True:
            left = z.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), 2 if last else 1)   # ZSTD_e_end / ZSTD_e_flush
            assert not z.ZSTD_isError(left)
            if left == 0 and ib.pos == ib.size:
                break
        at += n
    z.ZSTD_freeCCtx(cctx)
    return dst.raw[:ob.pos]
