#!/usr/bin/env python3
"""Soak of K1's two-level atomic epilogue: many unsynchronised launches whose workgroups all finish together (1-24 steps
per workgroup, the sizes that take the grouped path), on two streams adding to ONE counter array, checked against the
oracle every round.  A member's add that the group leader missed, or a copy not left at zero, shows up as a wrong total.
   python3 tests/perf/soak_epilogue.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from libflagstats_amd import _lib, device  # noqa: E402

lib = _lib.lib()
_lib.check(lib.FLAGSTATS_hip_init(0), "init")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
STEP = 16384
n = STEP * 256 * 24
t = torch.empty(n, dtype=torch.int16, device="cuda:0")
device.generate_torch(t, device.GEN_UNIFORM, seed=99, mask=0xFFFF)
host = oracle.generate(oracle.GEN_UNIFORM, 99, 0xFFFF, 0, n)
hist_cache = {}
rs = np.random.RandomState(1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.time()
launches = 0
for r in range(rounds):
    out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    want = np.zeros(32, dtype=np.uint64)
    for i in range(200):
        steps = int(rs.choice([64, 100, 256, 257, 300, 512, 1000, 256 * 3, 256 * 8 + 5, 256 * 24]))
        m = STEP * steps - int(rs.randint(0, 9))
        a = int(rs.randint(0, (n - m) // 8 + 1)) * 8
        with torch.cuda.stream(s1 if i & 1 else s2):
            device.count_torch(t[a:a + m], out)
        key = (a, m)
        if key not in hist_cache:
            hist_cache[key] = oracle.flagstat_hist(host[a:a + m])
        want += hist_cache[key]
        launches += 1
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want), (r, got, want)
    if r % 20 == 0:
        print("round %d ok (%d launches, %.0f s)" % (r, launches, time.time() - t0), flush=True)
print("soak ok: %d launches in %.0f s, all totals exact" % (launches, time.time() - t0))
