#!/usr/bin/env python3
"""Per-launch K1+K2 times from a cold start: shows how many launches the chip needs to reach its
steady state (clock / power management), i.e. what a short bench (5 warm-up + 20 steps) measures
compared with a long one (20 + 500).  One hipEvent between consecutive launches on the launch stream.

    python tools/step_times.py [--flags 4294967296] [--steps 120] [--idle-s 2.0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=2 ** 32)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--idle-s", type=float, default=2.0)
    args = ap.parse_args()

    import torch

    from libflagstats_amd import _lib, device

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.lib()
    _lib.check(lib.FLAGSTATS_hip_init(0), "init")
    flags = torch.empty(args.flags, dtype=torch.int16, device=dev)
    device.generate_torch(flags, device.GEN_UNIFORM, seed=2026, mask=0xFFFF)
    counters = torch.zeros(32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def series(tag):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        evs[0].record()
        for i in range(args.steps):
            device.count_torch(flags, counters)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        s = sorted(ms)
        print(json.dumps({"series": tag, "first10": [round(x, 4) for x in ms[:10]],
                          "steps_5_24_mean": round(sum(ms[5:25]) / 20, 4),
                          "last20_mean": round(sum(ms[-20:]) / 20, 4),
                          "median": round(s[len(s) // 2], 4), "min": round(s[0], 4), "max": round(s[-1], 4),
                          "every10": [round(x, 4) for x in ms[::10]]}), flush=True)

    series("cold (right after generation)")
    series("hot (immediately after)")
    time.sleep(args.idle_s)
    series("after %.1f s idle" % args.idle_s)
    time.sleep(10.0)
    series("after 10 s idle")


if __name__ == "__main__":
    main()
