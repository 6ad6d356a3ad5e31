#!/usr/bin/env python3
"""The GPU Zstandard decoder's two kernels called directly (fsk_zstd_decode) on frames made with the image's libzstd: decoded
bytes compared byte for byte, status codes, kernel time.  A development tool: the product path is the block-file entries
(tests/test_gpu_zstd.py)."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import blockfile_tool as bt  # noqa: E402
from libflagstats_amd import _lib  # noqa: E402


class GpuBlock(ctypes.Structure):
    _fields_ = [("src_off", ctypes.c_uint64), ("dst_off", ctypes.c_uint64), ("src_len", ctypes.c_uint32), ("dst_len", ctypes.c_uint32)]


def decode_frames(lib, frames, sizes, prof=0, reps=1, min_blocks=0):
    """frames: list of bytes, sizes: decoded size of each -> (list of decoded bytes, status array, tally, best kernel ms);
    min_blocks > 0: the second-pass entry (fsk_zstd_decode_ex: room for that many Zstandard blocks per frame)"""
    n = len(frames)
    blocks = (GpuBlock * n)()
    comp = bytearray(8)
    dpos = 0
    for i, (f, sz) in enumerate(zip(frames, sizes)):
        comp += b"\0" * 8
        blocks[i] = GpuBlock(len(comp), dpos, len(f), sz)
        comp += f
        dpos += (sz + 15) & ~15
    comp += b"\0" * 64
    d_comp = torch.frombuffer(comp, dtype=torch.uint8).cuda()
    d_blocks = torch.frombuffer(bytearray(bytes(blocks)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(dpos + 64, dtype=torch.uint8, device="cuda")
    d_status = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    d_tally = torch.zeros(32, dtype=torch.int64, device="cuda")
    lib.fsk_zstd_scratch_bytes_ex.restype = ctypes.c_uint64
    lib.fsk_zstd_scratch_bytes_ex.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    mx = max(sizes) if sizes else 0
    need = lib.fsk_zstd_scratch_bytes_ex(mx, n, min_blocks)
    d_scratch = torch.zeros(need + 256, dtype=torch.uint8, device="cuda")
    sp = (d_scratch.data_ptr() + 255) & ~255
    lib.fsk_zstd_decode_ex.restype = ctypes.c_int
    lib.fsk_zstd_decode_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    best = 1e9
    for _ in range(reps):
        d_status.fill_(-1)
        d_tally.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.fsk_zstd_decode_ex(d_comp.data_ptr(), d_blocks.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), d_tally.data_ptr(), sp, need, mx, min_blocks, prof, None)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        if rc:
            raise RuntimeError("fsk_zstd_decode: hip error %d" % rc)
    out = d_out.cpu().numpy()
    st = d_status.cpu().numpy()
    got = [bytes(out[blocks[i].dst_off:blocks[i].dst_off + sizes[i]]) for i in range(n)]
    return got, st, d_tally.cpu().numpy(), best


def cases(big):
    import oracle
    r = np.random.default_rng(5)
    yield "empty", b""
    yield "one", b"a"
    yield "abc", b"abc" * 5
    yield "zeros1000", bytes(1000)
    yield "zeros300k", bytes(300000)
    yield "random300", os.urandom(300)
    yield "random200k", os.urandom(200000)
    yield "hello", b"hello world, " * 3000
    f = oracle.generate(oracle.GEN_NA12878, 7, 1, 0, 700000).tobytes()
    yield "na12878_70k", f[:70000]
    yield "na12878_1024000", f[:1024000]
    yield "na12878_1.4M", f
    yield "u8_4", r.integers(0, 4, 300000, dtype=np.uint8).tobytes()
    yield "u8_60", r.integers(0, 60, 200000, dtype=np.uint8).tobytes()
    yield "u16_3000", r.integers(0, 3000, 100000, dtype=np.uint16).tobytes()
    a = os.urandom(40000)
    yield "longruns", a + bytes(50000) + a + os.urandom(20000) + a[:30000] + bytes(100000)
    yield "alternating", (os.urandom(17000) + b"x" * 17000) * 12
    if big:
        yield "uniform_u16", oracle.generate(oracle.GEN_UNIFORM, 3, 1, 0, 512000).tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="1,3,9,19,-5")
    ap.add_argument("--prof", type=int, default=0)
    ap.add_argument("--many", type=int, default=0, help="also time this many copies of the 1,024,000-byte NA12878-like frame")
    ap.add_argument("--only", default="")
    ap.add_argument("--many-kind", default="na12878", help="na12878 | uniform (12-bit uniform flags: no far matches)")
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    bad = 0
    for name, raw in cases(True):
        if args.only and args.only not in name:
            continue
        for level in [int(x) for x in args.levels.split(",")]:
            comp = bt.compress_block(raw, "zstd", level)
            got, st, tally, ms = decode_frames(lib, [comp], [len(raw)], args.prof)
            even = len(raw) & ~1   # (an odd last byte of a block is no flag: the decoders drop it, benchmark/flagstats.cpp:323)
            ok = st[0] == 0 and got[0][:even] == raw[:even]
            line = "%-18s level %3d: %7d -> %7d bytes, status %d, %s, %.2f ms, %d records, %d far" % (name, level, len(raw), len(comp), st[0], "exact" if ok else "WRONG", ms, tally[0], tally[1])
            if not ok:
                bad += 1
                if st[0] == 0:
                    g = np.frombuffer(got[0], dtype=np.uint8)
                    w = np.frombuffer(raw, dtype=np.uint8)
                    d = np.nonzero(g != w)[0]
                    line += " | %d bytes differ, first at %d (got %s want %s)" % (len(d), d[0], g[d[0]:d[0] + 8].tolist(), w[d[0]:d[0] + 8].tolist())
            print(line, flush=True)
    if args.many:
        import oracle
        per = 512000
        frames, sizes, raws = [], [], []
        for i in range(args.many):
            raw = (oracle.generate(oracle.GEN_UNIFORM, 3, 0x0FFF, i * per, per) if args.many_kind == "uniform" else oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, per)).tobytes()
            frames.append(bt.compress_block(raw, "zstd", 1))
            sizes.append(len(raw))
            raws.append(raw)
        got, st, tally, ms = decode_frames(lib, frames, sizes, args.prof, reps=3)
        nbad = sum(1 for i in range(args.many) if st[i] != 0 or got[i] != raws[i])
        if args.prof:
            nb = float(args.many)
            ne, ns = ctypes.c_int(0), ctypes.c_int(0)
            lib.fsk_zstd_role_waves(ctypes.byref(ne), ctypes.byref(ns))
            ne, ns = ne.value, ns.value
            print("cycles per frame: prepare %.3g (literals %.3g, tables %.3g) | chain %.3g per wave of two frames = %.0f per step (%.0f steps a frame) | records %.3g (%.1f relaxation rounds a batch, %.0f batches) | "
                  "emit %.3g x%d (waiting %.0f %%), scan %.3g x%d (waiting %.0f %%), copy %.3g (waiting %.0f %%) | %.0f batches, %.0f groups, %.0f chunks" % (
                tally[17] / nb, tally[18] / nb, tally[19] / nb, tally[20] / max(tally[21], 1), tally[20] / max(tally[21], 1) / max(tally[22] / nb / 8, 1), tally[22] / nb, tally[23] / nb,
                tally[24] / max(tally[25], 1), tally[25] / nb, tally[15] / nb / ne, ne, 100.0 * tally[16] / max(tally[15], 1), tally[8] / nb / ns, ns, 100.0 * tally[9] / max(tally[8], 1),
                tally[12] / nb, 100.0 * tally[13] / max(tally[12], 1), tally[5] / nb, tally[6] / nb, tally[14] / nb))
        if args.prof:
            print("chain, per wave of two frames: %.0f refills of the bit-stream rings, %.0f cycles each = %.1f %% of the wave's cycles" % (
                tally[27] / max(tally[21], 1), tally[26] / max(tally[27], 1), 100.0 * tally[26] / max(tally[20], 1)))
        if args.prof:
            print("emitters, per frame: %.0f short literal runs (a 4-byte store), %.0f long ones, %.0f groups with far matches in %.0f rounds of 16 bytes; cycles per batch and emitter: "
                  "loads, scans and checks %.0f | markers and short runs %.0f | long runs %.0f | far matches %.0f" % (
                tally[7] / nb, tally[2] / nb, tally[28] / nb, tally[4] / nb, tally[30] / max(tally[5], 1), tally[31] / max(tally[5], 1), tally[3] / max(tally[5], 1), tally[29] / max(tally[5], 1)), flush=True)
        print("%d frames of 1,024,000 bytes (zstd-1): %d wrong, best %.2f ms = %.1f Gflags/s; %d records, %d far" % (args.many, nbad, ms, args.many * per / ms / 1e6, tally[0], tally[1]), flush=True)
        bad += nbad
    print("FAILED: %d" % bad if bad else "all exact")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
