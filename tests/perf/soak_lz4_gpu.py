#!/usr/bin/env python3
"""Soak of the GPU LZ4 (--mode fast:2 | hc:9) or Zstandard (--mode zstd:1) decode pipeline (copy stream + two decode streams + events, file and image mode): the same image
through the product entry again and again, alternating with the host-thread decoder, counters checked every time.
Every round is printed with its stages (copies, decode left behind the last copy) and with what the job's CPU quota did during
it (cgroup v2 cpu.stat: throttled periods and time): a GPU box gives a 1-GPU job 16 CPUs' worth of time per 100 ms period, the
host-thread rounds (20 decoder threads) and the 16 file readers spend it, and a round that runs into an exhausted period waits
for the next one -- which is where r04's unexplained 36.6-57.5 ms spread came from (--host-every 0 leaves the host rounds out)."""
import argparse
import ctypes
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import numpy as np  # noqa: E402

from libflagstats_amd import _lib  # noqa: E402
from lz4_decoder_sweep import build_image  # noqa: E402


def cpu_stat():
    """(periods, throttled periods, throttled ms) of this job's cgroup, zeros if unreadable"""
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return int(d.get("nr_periods", 0)), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e3
    except OSError:
        return 0, 0, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 31)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--mode", default="fast:2")
    ap.add_argument("--host-every", type=int, default=5, help="every n-th round decodes on the host threads (0: never)")
    ap.add_argument("--gap-ms", type=float, default=0.0, help="sleep between rounds")
    args = ap.parse_args()
    import oracle
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    mode, level = args.mode.split(":")
    zstd = mode == "zstd"
    knob = b"zstd_decoder" if zstd else b"lz4_decoder"
    file_entry = lib.FLAGSTATS_hip_blockfile_zstd if zstd else lib.FLAGSTATS_hip_blockfile_lz4
    image_entry = lib.FLAGSTATS_hip_blockimage_zstd if zstd else lib.FLAGSTATS_hip_blockimage_lz4
    img = build_image(args.flags, mode, int(level))
    want = oracle.flagstat_generated(oracle.GEN_NA12878, 7, 1, 0, args.flags)
    buf = np.frombuffer(img, dtype=np.uint8)
    walls = {0: [], 1: []}
    rows = []
    with tempfile.NamedTemporaryFile(suffix=".zst" if zstd else ".lz4", dir=os.environ.get("TMPDIR", "/tmp")) as f:
        f.write(img)
        f.flush()
        os.fsync(f.fileno())   # written back before the timed reads: a file in the page cache, clean
        for r in range(args.rounds):
            dec = 0 if args.host_every and r % args.host_every == args.host_every - 1 else 1
            if args.gap_ms:
                time.sleep(args.gap_ms * 1e-3)
            c0 = cpu_stat()
            _lib.check(lib.FLAGSTATS_hip_set(knob, dec), "set")
            out = np.zeros(32, dtype=np.uint64)
            st = _lib.BlockfileStats()
            t0 = time.perf_counter()
            if r % 2:
                _lib.check(file_entry(f.name.encode(), 0, out.ctypes.data, ctypes.byref(st)), "blockfile")
            else:
                _lib.check(image_entry(buf.ctypes.data, buf.size, 0, out.ctypes.data, ctypes.byref(st)), "blockimage")
            walls[dec].append(time.perf_counter() - t0)
            c1 = cpu_stat()
            rows.append((r, "GPU" if dec else "host", "file" if r % 2 else "image", walls[dec][-1] * 1e3, st.wait_copy_s * 1e3, st.wait_decode_s * 1e3, c1[1] - c0[1], c1[2] - c0[2]))
            assert st.gpu_decode == dec
            assert np.array_equal(out, want), "round %d (%s, %s): counters differ from the oracle" % (r, "GPU" if dec else "host", "file" if r % 2 else "image")
    _lib.check(lib.FLAGSTATS_hip_set(knob, 2), "set")
    g = walls[1]
    print("  round decoder mode   wall ms  copies ms  decode behind the last copy ms  throttled periods  throttled ms (summed over the cgroup's threads)")
    for row in rows:
        print("  %5d %-7s %-6s %7.1f  %9.1f  %30.1f  %17d  %12.1f" % row)
    steady = sorted(g[1:])
    if steady:
        med = steady[len(steady) // 2]
        print("  GPU decode, steady state: median %.1f ms, max %.1f ms = %.2f x the median" % (med * 1e3, steady[-1] * 1e3, steady[-1] / med))
        for what in ("image", "file"):
            # (the first GPU round of a mode makes that mode's own resources -- file mode: the page-locked ring -- and is left out)
            w = sorted(r[3] for r in [r for r in rows if r[1] == "GPU" and r[2] == what][1:])
            if w:
                print("    %s mode alone, its first round left out: median %.1f ms, max %.1f ms = %.2f x" % (what, w[len(w) // 2], w[-1], w[-1] / w[len(w) // 2]))
    print("soak: %d rounds on %d flags (%s), image and file mode alternating, all exact" % (args.rounds, args.flags, args.mode))
    print("  GPU decode, every round in order (ms; even = image, odd = file): " + " ".join("%.1f" % (w * 1e3) for w in g))
    print("  GPU decode: FIRST call %.1f ms (it allocates the large device buffers, the pinned spans, streams and events), afterwards %.1f-%.1f ms; host threads %.1f-%.1f ms"
          % (g[0] * 1e3, min(g[1:]) * 1e3, max(g[1:]) * 1e3, min(walls[0] or [0]) * 1e3, max(walls[0] or [0]) * 1e3))


if __name__ == "__main__":
    main()
