#!/usr/bin/env python3
"""Why does a branch-free kernel run 1-3 % slower on NA12878-like flags than on uniform-random ones?

Two modes.
  (default)   A/B in ONE process: one buffer per input kind (uniform 0xFFFF, NA12878-like, uniform 0x0FFF, all zero),
              launches interleaved kind by kind for many rounds, median / best per kind, with the GPU's clocks and
              power sampled from sysfs (pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk, hwmon power) while each kind runs.
  --single K  only kind K, `--launches` back-to-back launches: the program to put behind `rocprofv3 --pmc ...`
              (one counter set per pass, tools/kind_compare.sh), so that the counters of the two kinds can be
              compared launch for launch.
"""
import argparse
import glob
import os
import statistics
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libflagstats_amd import _lib, device  # noqa: E402

KINDS = {0: ("uniform 0xFFFF", 0, 0xFFFF), 1: ("NA12878-like", 1, 1), 2: ("uniform 0x0FFF", 0, 0x0FFF), 3: ("all zero", 0, 0)}


def star(path):
    """current level of a pp_dpm_* file: the line marked with '*' -> MHz"""
    try:
        for line in open(path):
            if "*" in line:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except (OSError, ValueError, IndexError):
        pass
    return None


class Sampler:
    """GPU clocks and power from sysfs (readable without privileges where the box exposes them)"""

    def __init__(self):
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.base = os.path.dirname(cards[0]) if cards else None
        hw = glob.glob(os.path.join(self.base, "hwmon", "hwmon*")) if self.base else []
        self.hw = hw[0] if hw else None
        self.rows = []

    def read(self):
        if not self.base:
            return None
        r = {k: star(os.path.join(self.base, "pp_dpm_" + k)) for k in ("sclk", "mclk", "fclk")}
        for name in ("power1_average", "power1_input"):
            try:
                r["W"] = int(open(os.path.join(self.hw, name)).read()) / 1e6
                break
            except (OSError, ValueError, TypeError):
                pass
        try:
            r["freq1"] = int(open(os.path.join(self.hw, "freq1_input")).read()) / 1e6
        except (OSError, ValueError, TypeError):
            pass
        return r

    def sample_until(self, stop):
        while not stop.is_set():
            row = self.read()
            if row:
                self.rows.append(row)
            time.sleep(0.003)

    def summary(self):
        out = []
        for k in ("sclk", "freq1", "mclk", "fclk", "W"):
            v = [r[k] for r in self.rows if r.get(k) is not None]
            if v:
                out.append("%s %.0f (min %.0f max %.0f)" % (k, statistics.mean(v), min(v), max(v)))
        return ", ".join(out) if out else "no sysfs clock files readable"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 32)
    ap.add_argument("--kinds", default="0,1,2,3")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--reps", type=int, default=25)
    ap.add_argument("--single", type=int, default=None)
    ap.add_argument("--launches", type=int, default=30)
    args = ap.parse_args()
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    n = args.flags
    if args.single is not None:
        name, kind, mask = KINDS[args.single]
        d = device.DeviceFlags(n).generate(kind, seed=5, mask=mask)
        ms, _ = device.time_device_ptr(d.ptr, n, 10, args.launches)
        print("%-16s n=%d: %d launches, %.3f us per launch, %.3f TB/s" % (name, n, args.launches, ms / args.launches * 1e3,
                                                                        2 * n / (ms / args.launches) / 1e9))
        return
    kinds = [int(k) for k in args.kinds.split(",")]
    bufs = {k: device.DeviceFlags(n).generate(KINDS[k][1], seed=5, mask=KINDS[k][2]) for k in kinds}
    res = {k: [] for k in kinds}
    clk = {k: Sampler() for k in kinds}
    for k in kinds:   # settle
        device.time_device_ptr(bufs[k].ptr, n, 5, 20)
    for r in range(args.rounds):
        order = kinds if r % 2 == 0 else kinds[::-1]
        for k in order:
            stop = threading.Event()
            th = threading.Thread(target=clk[k].sample_until, args=(stop,), daemon=True)
            th.start()
            ms, _ = device.time_device_ptr(bufs[k].ptr, n, 2, args.reps)
            stop.set()
            th.join()
            res[k].append(ms / args.reps)
    print("n = %d flags (%.0f MiB), %d rounds x %d launches per kind, kinds interleaved (order reversed every other round)"
          % (n, n * 2 / 2 ** 20, args.rounds, args.reps))
    base = statistics.median(res[kinds[0]])
    for k in kinds:
        t = res[k]
        med = statistics.median(t)
        print("  %-16s median %9.2f us %6.3f TB/s (%+.2f %% vs %s)   best %9.2f   worst %9.2f   | %s"
              % (KINDS[k][0], med * 1e3, 2 * n / med / 1e9, (med / base - 1) * 100, KINDS[kinds[0]][0], min(t) * 1e3, max(t) * 1e3,
                 clk[k].summary()))


if __name__ == "__main__":
    main()
