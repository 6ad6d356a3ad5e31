#!/usr/bin/env python3
"""VERDICT r01 item 5, second half: does a captured HIP graph of the K1 + K2 pair beat launching it?
Captures one store-form call (K1 + K2) and one accumulate call (K1 with the atomic epilogue) on a torch
stream and compares replays with plain launches, per array size (event-timed, back to back)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
dev = torch.device("cuda", 0)
sizes = [10 ** 6, 2 ** 22, 2 ** 26, 2 ** 29]
buf = torch.empty(max(sizes), dtype=torch.int16, device=dev)
device.generate_torch(buf, device.GEN_UNIFORM, seed=5, mask=0xFFFF)
out = torch.zeros(32, dtype=torch.int64, device=dev)
s = torch.cuda.Stream(device=dev)
print("%-12s %-22s %12s %12s" % ("flags", "form", "launch us", "graph us"))
for n in sizes:
    t = buf[:n]
    for form, store in (("K1+K2 (store)", True), ("K1 atomic (+=)", False)):
        with torch.cuda.stream(s):
            for _ in range(3):
                device.count_torch(t, out, store=store)      # workspace exists before capture
            s.synchronize()
            reps = 200

            def timed(fn):
                best = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(reps):
                        fn()
                    e1.record(s)
                    s.synchronize()
                    best.append(e0.elapsed_time(e1) / reps * 1e3)
                return statistics.median(best)

            plain = timed(lambda: device.count_torch(t, out, store=store))
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    device.count_torch(t, out, store=store)
                graph = timed(g.replay)
            except Exception as e:  # noqa: BLE001
                graph = float("nan")
                print("capture failed:", repr(e)[:200])
        print("%-12d %-22s %12.2f %12.2f" % (n, form, plain, graph), flush=True)
