#!/usr/bin/env python3
"""One-shot times: what a user of the reference's one-shot programs sees.  The reference publishes process-level times
from a file (README.md:31-36: 0.72 s for the LZ4-HC-c9 block file, 0.48 s for the raw file of its 824,541,892-flag readset;
benchmark/flagstats.cpp:288-342 is a program that starts, reads, counts, prints and exits), so the number to put beside
them is wall time from process start to counters in a FRESH process -- hipInit, code-object load, engine creation, the
decoder's buffers and all -- not the best of repeated calls in a warm process.

This script (the parent) never touches the GPU: it writes the input files, then starts a fresh child per sample
(tests/perf/oneshot.c, linked against libflagstats_hip.so) and measures from just before the spawn to the child's own
"counters ready" stamp on the system-wide monotonic clock; the child repeats the call twice more, which gives the warm
figure of the same process for comparison, and prints the library's host-side phases (FLAGSTATS_HIP_GPU_DECODE_TIMES).
The reference's own program (oracle/_ref/bench_ref = benchmark/flagstats.cpp unmodified, CPU) reads the same files on
the same box for scale.  Files are in the page cache (they were just written; a non-root job cannot drop it)."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
HERE = os.path.dirname(os.path.abspath(__file__))
ONESHOT = os.path.join(HERE, "build", "oneshot")
BENCH_REF = os.path.join(ROOT, "oracle", "_ref", "bench_ref")


def build_oneshot():
    src = os.path.join(HERE, "oneshot.c")
    if os.path.exists(ONESHOT) and os.path.getmtime(ONESHOT) >= os.path.getmtime(src):
        return
    os.makedirs(os.path.dirname(ONESHOT), exist_ok=True)
    libdir = os.path.join(ROOT, "libflagstats_amd")
    subprocess.run(["gcc", "-O2", "-o", ONESHOT, src, "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-l:libflagstats_hip.so",
                    "-Wl,-rpath," + libdir], check=True)


EVICT = False


def evict(path):
    """--evict: the file's (clean, written-back) pages are dropped from the page cache before a sample: a job without root
    cannot drop the whole cache, but POSIX_FADV_DONTNEED on its own file does that much"""
    if EVICT and os.path.isfile(str(path)):
        fd = os.open(path, os.O_RDONLY)
        try:
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)


GAP_S = 1.0


CALLS = 3


def spawn_oneshot(mode, arg, calls=None, env_extra=None):
    calls = CALLS if calls is None else calls
    env = dict(os.environ, FLAGSTATS_HIP_GPU_DECODE_TIMES="1", FLAGSTATS_HIP_INIT_TIMES="1")
    env.update(env_extra or {})
    evict(arg)
    time.sleep(GAP_S)   # (the process before has exited; the driver releases its device memory behind it -- a one-shot user does not start in that wake)
    t_spawn = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    r = subprocess.run([ONESHOT, str(t_spawn), mode, str(arg), str(calls)], capture_output=True, text=True, env=env)
    t_exit = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    if r.returncode:
        raise RuntimeError("oneshot %s %s failed: %s" % (mode, arg, r.stderr[-2000:]))
    d = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            d.update(json.loads(ln))
    d["process_ms"] = (t_exit - t_spawn) * 1e-6
    d["teardown_ms"] = d["process_ms"] - d.get("main_returns_since_spawn_ms", d["process_ms"])   # return from main -> the parent sees it gone
    d["shutdown"] = [ln for ln in r.stderr.splitlines() if ln.startswith("oneshot: FLAGSTATS_hip_shutdown")]
    d["phases"] = [ln for ln in r.stderr.splitlines() if ln.startswith("gpu decode, host side")]
    d["ring"] = [ln for ln in r.stderr.splitlines() if ln.startswith("gpu decode, pinned ring") or ln.startswith("gpu decode, presets")]
    d["init_phases"] = [ln for ln in r.stderr.splitlines() if ln.startswith("engine creation")]
    return d


def med(xs):
    return statistics.median(xs)


def report(name, samples, n_flags, ref_note=""):
    ready = [s["counters_ready_since_spawn_ms"] for s in samples]
    first = [s["calls_ms"][0] for s in samples]
    warm = [min(s["calls_ms"][1:]) for s in samples if len(s["calls_ms"]) > 1]
    line = ("%-34s start -> counters %7.1f ms (min %.1f, max %.1f; %d fresh processes) = exec + loader %.1f + input %.1f + hipInit / engine %.1f + "
            "FIRST call %.1f" % (name, med(ready), min(ready), max(ready), len(samples), med([s["since_spawn_at_main_ms"] for s in samples]),
                                 med([s["input_ms"] for s in samples]), med([s["init_ms"] for s in samples]), med(first)))
    if warm:
        line += " | the same call again in that process %.1f ms: first / warm = %.2fx" % (med(warm), med(first) / med(warm))
    line += " | whole process (%d call%s) %.1f ms, of which return from main -> process gone %.1f | %.2f Gflags/s one-shot" % (
        len(samples[0]["calls_ms"]), "" if len(samples[0]["calls_ms"]) == 1 else "s", med([s["process_ms"] for s in samples]), med([s["teardown_ms"] for s in samples]),
        n_flags / med(ready) / 1e6)
    print(line + ref_note, flush=True)
    print("    every sample: start -> counters %s ms; first call %s ms" % (" ".join("%.0f" % x for x in ready), " ".join("%.0f" % x for x in first)), flush=True)
    typical = min(samples, key=lambda s: abs(s["calls_ms"][0] - med(first)))
    slowest = max(samples, key=lambda s: s["calls_ms"][0])
    if typical.get("shutdown"):
        print("    " + typical["shutdown"][0], flush=True)
    if typical.get("init_phases"):
        print("    " + typical["init_phases"][0], flush=True)
    if typical["phases"]:
        print("    first call (the median sample), " + typical["phases"][0], flush=True)
        for ln in typical.get("ring", [])[:2]:
            print("    " + ln, flush=True)
        if len(typical["phases"]) > 1:
            print("    second call, " + typical["phases"][1], flush=True)
    if slowest is not typical and slowest["calls_ms"][0] > 1.5 * med(first) and slowest["phases"]:
        print("    first call (the SLOWEST sample), " + slowest["phases"][0], flush=True)


def reference_program(path, raw, evict_path=None):
    """the reference's own binary on the same file (CPU, one thread): wall time of the whole process, best of 2"""
    if not os.path.exists(BENCH_REF):
        return None
    best = None
    for _ in range(2):
        evict(evict_path or path)
        t0 = time.perf_counter()
        r = subprocess.run([BENCH_REF, "decompress", "-i", path, "-D" if raw else "-d"], capture_output=True, text=True)
        t = (time.perf_counter() - t0) * 1e3
        if r.returncode == 0:
            best = t if best is None else min(best, t)
    return best


def python_oneshot(n, env_extra=None):
    """python/README.md:45-47: `fs.flagstats(np.random.randint(0, 8192, 100_000_000, dtype="uint16"))` "completes in around 1 second" """
    code = ("import time, sys; t0 = time.perf_counter(); import numpy as np; t1 = time.perf_counter(); sys.path.insert(0, %r); import pyflagstats as fs; t2 = time.perf_counter();"
            "a = np.random.randint(0, 8192, %d, dtype='uint16'); t3 = time.perf_counter(); r = fs.flagstats(a); t4 = time.perf_counter(); r2 = fs.flagstats(a); t5 = time.perf_counter();"
            "print('PY %%.1f %%.1f %%.1f %%.1f %%.1f %%d' %% ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, r['n_values']))") % (ROOT, n)
    time.sleep(GAP_S)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **(env_extra or {})))
    wall = (time.perf_counter() - t0) * 1e3
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    f = [ln for ln in r.stdout.splitlines() if ln.startswith("PY ")][-1].split()
    return wall, [float(x) for x in f[1:6]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=824541892, help="flags per file (default: the README's readset)")
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--which", default="u16,python,hc9,fast,zstd,raw")
    ap.add_argument("--env", default="", help="KEY=VALUE[,KEY=VALUE] for the children (A/B of library knobs)")
    ap.add_argument("--calls", type=int, default=3, help="calls per fresh process (1: the one-shot program itself; the whole-process time is then the reference's measure)")
    ap.add_argument("--gap-s", type=float, default=1.0, help="pause before every fresh process")
    ap.add_argument("--lazy-init", action="store_true", help="the children do not call FLAGSTATS_hip_init: the first call creates the engine (init_ms then reads 0 and is inside the first call)")
    ap.add_argument("--evict", action="store_true", help="drop the file from the page cache before every sample (the FIRST call then reads the disk; the repeated calls of the same process are warm again)")
    args = ap.parse_args()
    global EVICT, GAP_S, CALLS
    CALLS = args.calls
    EVICT = args.evict
    GAP_S = args.gap_s
    which = args.which.split(",")
    env_extra = dict(kv.split("=", 1) for kv in args.env.split(",") if kv)
    if args.lazy_init:
        env_extra["ONESHOT_LAZY_INIT"] = "1"
    build_oneshot()
    tmp = os.environ.get("TMPDIR", "/tmp")
    print("one-shot times, %d fresh processes per line; files of %d flags in %s (%s)%s" % (args.samples, args.flags, tmp, "EVICTED from the page cache before every sample" if EVICT else "page cache", " env " + args.env if args.env else ""), flush=True)
    if "u16" in which:
        for n in (1_000_000, 100_000_000):
            report("FLAGSTATS_u16, %d flags" % n, [spawn_oneshot("u16", n, env_extra=env_extra) for _ in range(args.samples)], n)
    if "python" in which:
        # (the package loads torch's bundled HIP runtime first when torch is installed, so that a later `import torch` in the same
        # process binds to the same copy: libflagstats_amd/_lib.py; FLAGSTATS_HIP_SYSTEM_RUNTIME=1 skips that)
        for label, extra in (("pyflagstats.flagstats, 1e8 flags", {}), ("... FLAGSTATS_HIP_SYSTEM_RUNTIME=1", {"FLAGSTATS_HIP_SYSTEM_RUNTIME": "1"})):
            rows = [python_oneshot(100_000_000, extra) for _ in range(max(2, args.samples // 2))]
            wall = med([w for w, _ in rows])
            parts = [med([p[i] for _, p in rows]) for i in range(5)]
            print("%-34s whole interpreter run %7.1f ms = import numpy %.1f + import pyflagstats (no GPU call yet) %.1f + np.random.randint %.1f + FIRST flagstats() %.1f (loading the HIP runtime, "
                  "hipInit, engine, 200 MB over PCIe) | second flagstats() %.1f ms  [python/README.md:45-47: \"around 1 second\" incl. RNG]" % (label, wall, *parts), flush=True)
    if any(k in which for k in ("hc9", "fast", "zstd", "raw")):
        import numpy as np  # noqa: F401

        import oracle
        from lz4_decoder_sweep import build_image
        for key, mode, level, suffix, readme in (("hc9", "hc", 9, ".lz4", "README.md:35: 0.72 s"), ("fast", "fast", 2, ".lz4", "README.md:155: 0.990 s (LZ4-fast-c2)"),
                                                 ("zstd", "zstd", 1, ".zst", "README.md:157: 3.630 s (Zstd-c1)")):
            if key not in which:
                continue
            img = build_image(args.flags, mode, level)
            with tempfile.NamedTemporaryFile(suffix=suffix, dir=tmp) as f:
                f.write(img)
                f.flush()
                os.fsync(f.fileno())   # written back before the timed reads: a file in the page cache, clean
                size = len(img)
                del img
                samples = [spawn_oneshot("blockfile", f.name, env_extra=env_extra) for _ in range(args.samples)]
                assert all(s["n_flags"] == args.flags and s["gpu_decode"] == 1 and s["checksum"] == samples[0]["checksum"] for s in samples)
                ref = reference_program(f.name, raw=False)
                note = " | reference program on this host, same file: %.0f ms [%s, hardware not stated]" % (ref, readme) if ref else " [%s]" % readme
                report("%s block file, %.0f MiB" % ({"hc9": "LZ4-HC-c9", "fast": "LZ4-fast-c2", "zstd": "Zstd-c1"}[key], size / 2**20), samples, args.flags, note)
        if "raw" in which:
            per = 1 << 26
            # (the reference's `decompress` refuses an input name that does not end in .lz4 / .zst even for its raw modes -R / -D
            # (benchmark/flagstats.cpp:889-896), and its raw reader then opens the name cut behind ".bin" (:416-424): the file is
            # X.bin, the reference is given "X.bin.lz4")
            with tempfile.NamedTemporaryFile(suffix=".bin", dir=tmp) as f:
                for at in range(0, args.flags, per):
                    f.write(oracle.generate(oracle.GEN_NA12878, 7, 1, at, min(per, args.flags - at)).tobytes())
                f.flush()
                os.fsync(f.fileno())   # written back before the timed reads: a file in the page cache, clean
                samples = [spawn_oneshot("raw", f.name, env_extra=env_extra) for _ in range(args.samples)]
                assert all(s["n_flags"] == args.flags for s in samples)
                ref = reference_program(f.name + ".lz4", raw=True, evict_path=f.name)
                note = " | reference program on this host, same file: %.0f ms [README.md:36: 0.48 s, hardware not stated]" % ref if ref else " [README.md:36: 0.48 s]"
                report("raw uint16 file, %.0f MiB" % (2 * args.flags / 2**20), samples, args.flags, note)


if __name__ == "__main__":
    main()
