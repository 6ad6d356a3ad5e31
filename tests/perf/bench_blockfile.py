#!/usr/bin/env python3
"""BASELINE config 4: LZ4 512k-record block file -> counters, end to end on one MI355X host.

Builds an NA12878-like FLAG stream (the README's 824.5 M reads by default would take a while to
compress in Python; default here is 2^29 flags = 1 GiB), writes it in the reference's block format
with the image's liblz4, then times
  * the product: FLAGSTATS_hip_blockfile_lz4 at several thread counts (file in page cache), and
  * the reference's own loop shape on this host: liblz4 LZ4_decompress_safe + the reference's
    dispatcher kernel per block, serially on one thread (benchmark/flagstats.cpp:311-332),
    plus decode-only, as in the README's "decomp" / "flagstat" columns (README.md:136-175).
"""
import argparse
import ctypes
import json
import os
import struct
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import blockfile_tool as bt  # noqa: E402
from libflagstats_amd import _lib, blockfile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 29)
    ap.add_argument("--threads", default="1,4,16,32,64,128")
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--chunk-mib", default="64")
    ap.add_argument("--image", action="store_true", help="also time FLAGSTATS_hip_blockimage_lz4 on the file read into memory "
                                                         "(no pread of the compressed payload in the workers)")
    ap.add_argument("--gpu-decode", action="store_true", help="add a row per mode with the LZ4 blocks decoded on the GPU")
    ap.add_argument("--no-serial", action="store_true", help="skip the serial host loops of the reference shape")
    args = ap.parse_args()
    import oracle

    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    n = args.flags
    per = bt.BLOCK_BYTES // 2
    nblocks = (n + per - 1) // per

    def make(i):
        f = oracle.generate(oracle.GEN_NA12878, 7, 1, i * per, min(per, n - i * per))
        comp = bt.compress_block(f.tobytes(), args.mode, args.level)
        return struct.pack("<ii", f.nbytes, len(comp)) + comp

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(make, range(nblocks)))
    d = args.dir or tempfile.mkdtemp(prefix="fsblk_", dir="/tmp")
    path = os.path.join(d, "na12878_%s%d.%s" % (args.mode, args.level, "zst" if args.mode == "zstd" else "lz4"))
    with open(path, "wb") as f:
        for p in parts:
            f.write(p)
    size = os.path.getsize(path)
    print("built %s: %d flags, %d blocks, %d -> %d bytes (ratio %.2f) in %.1f s" %
          (path, n, nblocks, 2 * n, size, 2 * n / size, time.perf_counter() - t0), flush=True)
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, n)

    rows = []
    image = np.fromfile(path, dtype=np.uint8) if args.image else None
    modes = ["file"] + (["image"] if args.image else [])
    # this script measures the HOST-THREAD decode pipeline (thread counts, chunk sizes); --gpu-decode adds one row per mode
    # with the blocks decoded on the GPU (what the entries do by default for files of 1 GiB and more)
    lib.FLAGSTATS_hip_set(b"lz4_decoder", 0)
    sweep = [(m, int(c), int(t), 0) for m in modes for c in args.chunk_mib.split(",") for t in args.threads.split(",")]
    if args.gpu_decode and args.mode != "zstd":
        sweep += [(m, int(args.chunk_mib.split(",")[0]), 0, 1) for m in modes]
    for mode, cm, th, dec in sweep:
        lib.FLAGSTATS_hip_set(b"chunk_flags", cm << 19)
        lib.FLAGSTATS_hip_set(b"lz4_decoder", dec)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            if mode == "file":
                got, st = blockfile.flagstat_file(path, th)          # codec from the extension
            else:
                got = np.zeros(32, dtype=np.uint64)
                stc = _lib.BlockfileStats()
                entry = lib.FLAGSTATS_hip_blockimage_zstd if args.mode == "zstd" else lib.FLAGSTATS_hip_blockimage_lz4
                _lib.check(entry(image.ctypes.data, image.size, th, got.ctypes.data, ctypes.byref(stc)), "FLAGSTATS_hip_blockimage")
                st = {name: getattr(stc, name) for name, _ in stc._fields_}
            dt = time.perf_counter() - t0
            assert np.array_equal(got, want), "PARITY"
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        rows.append({"mode": mode, "decoder": "gpu" if st.get("gpu_decode") else "host threads", "chunk_MiB": cm, "threads": st["threads"], "wall_s": round(dt, 4), "Gflags_s": round(n / dt / 1e9, 2),
                     "compressed_GB_s": round(size / dt / 1e9, 2), "decode_cpu_s": round(st["decode_cpu_s"], 3), "setup_s": round(st["setup_s"], 4), "wait_decode_s": round(st["wait_decode_s"], 4), "wait_copy_s": round(st["wait_copy_s"], 4),
                     "decode_GB_s_per_thread": round(2 * n / max(st["decode_cpu_s"], 1e-9) / 1e9, 2)})
        print(rows[-1], flush=True)

    lib.FLAGSTATS_hip_set(b"lz4_decoder", 2)
    if args.no_serial or args.mode == "zstd":
        print(json.dumps({"workload": "%d NA12878-like flags, %d-byte LZ4-%s-%d blocks" % (n, bt.BLOCK_BYTES, args.mode, args.level),
                          "file_bytes": size, "product": rows}))
        os.remove(path)
        return
    # the reference's loop shape on this host: one thread, per block liblz4 decode then its dispatcher kernel
    ref = oracle.load_ref()
    img = open(path, "rb").read()
    base = ctypes.cast(ctypes.c_char_p(img), ctypes.c_void_p).value
    outbuf = ctypes.create_string_buffer(bt.BLOCK_BYTES + 65536)
    lz = bt.lz4()
    res = {}
    for what in ("decode_only", "decode_plus_reference_kernel"):
        if what != "decode_only" and ref is None:
            continue
        t0 = time.perf_counter()
        pos = 0
        counters = np.zeros(32, dtype=np.uint32)
        while pos < len(img):
            us, cs = struct.unpack_from("<ii", img, pos)
            pos += 8
            r = lz.LZ4_decompress_safe(base + pos, outbuf, cs, us)
            assert r == us
            if what != "decode_only":
                ref.ref_FLAGSTATS_u16(ctypes.cast(outbuf, ctypes.POINTER(ctypes.c_uint16)), us >> 1,
                                      counters.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
            pos += cs
        res[what] = time.perf_counter() - t0
    # the same serial caller loop, kernel call replaced by a streaming session (acquire -> liblz4
    # decodes straight into pinned memory -> commit): counting happens behind the decode thread
    from libflagstats_amd.session import StreamSession
    with StreamSession() as sess:
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            pos = 0
            while pos < len(img):
                us, cs = struct.unpack_from("<ii", img, pos)
                pos += 8
                p = sess.acquire_ptr((us >> 1) + 1)
                r = lz.LZ4_decompress_safe(base + pos, p, cs, us)
                assert r == us
                sess.commit(us >> 1)
                pos += cs
            got = sess.finish()
            dt = time.perf_counter() - t0
            assert np.array_equal(got, want), "PARITY (session)"
            best = dt if best is None else min(best, dt)
    res["decode_into_session_1_thread"] = best
    print(json.dumps({"workload": "%d NA12878-like flags, %d-byte LZ4-%s-%d blocks" % (n, bt.BLOCK_BYTES, args.mode, args.level),
                      "file_bytes": size, "product": rows,
                      "host_serial_reference_shape": {k: {"s": round(v, 4), "Gflags_s": round(n / v / 1e9, 3)}
                                                      for k, v in res.items()}}))
    os.remove(path)


if __name__ == "__main__":
    main()
