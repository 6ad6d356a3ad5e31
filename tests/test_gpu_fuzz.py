"""GPU: randomized parity sweep -- random lengths, 2-byte offsets, input makers and launch
geometries through the device entry, each checked bit-exactly against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_lengths_offsets_geometries(hip):
    import oracle
    from libflagstats_amd import _lib, device
    rs = np.random.RandomState(20261003)
    cap = 3_000_000
    buf = device.DeviceFlags(cap + 64)
    old_v, old_b = hip.FLAGSTATS_hip_get(b"variant"), hip.FLAGSTATS_hip_get(b"blocks_per_cu")
    old_f, old_e = hip.FLAGSTATS_hip_get(b"fuse"), hip.FLAGSTATS_hip_get(b"epilogue")
    # the shipped library carries the default schedule and the plain loop; a tuning build (make TUNING=1) all of them
    tuning = bool(hip.FLAGSTATS_hip_get(b"tuning_build"))
    variants = [9, 25, 71] + ([0, 1, 13, 17, 27, 29, 41, 61, 63, 65, 67, 69, 75, 77, 89, 153] if tuning else [])
    try:
        for it in range(int(os.environ.get("FLAGSTATS_FUZZ_ITERS", "150"))):   # a soak run sets thousands
            kind = int(rs.randint(0, 3))
            mask = [0xFFFF, 0x0FFF, 0x00FF][rs.randint(0, 3)] if kind == 0 else int(rs.randint(0, 2))
            seed = int(rs.randint(0, 2 ** 31))
            first = int(rs.randint(0, 2 ** 40))
            # lengths cluster around step (16384) and vector (8) boundaries as well as anywhere
            pick = rs.randint(0, 4)
            if pick == 0:
                n = int(rs.randint(0, 64))
            elif pick == 1:
                n = int(16384 * rs.randint(0, 40) + rs.randint(-9, 10))
            elif pick == 2:
                n = int(8 * rs.randint(0, 5000) + rs.randint(-1, 2))
            else:
                n = int(rs.randint(0, cap))
            n = max(0, min(n, cap))
            off = int(rs.randint(0, 32))
            _lib.check(hip.FLAGSTATS_hip_set(b"variant", int(rs.choice(variants))), "variant")
            _lib.check(hip.FLAGSTATS_hip_set(b"blocks_per_cu", int(rs.choice([1, 2, 3]))), "bpc")
            fuse = int(rs.randint(0, 2))                     # drawn in every build so the case sequence is the same
            if tuning:
                _lib.check(hip.FLAGSTATS_hip_set(b"fuse", fuse), "fuse")  # K1+K2 or the ticket-fused K1 (tuning build only)
            _lib.check(hip.FLAGSTATS_hip_set(b"epilogue", int(rs.randint(0, 2))), "epilogue")  # atomic adds from K1 or K2
            buf.generate(kind, seed=seed, mask=mask, first_index=first, offset=off, n=n)
            got = buf.count(offset=off, n=n)
            want = oracle.flagstat_generated(kind, seed, mask, first, n, threads=4)
            assert np.array_equal(got, want), (it, kind, mask, seed, first, n, off)
    finally:
        hip.FLAGSTATS_hip_set(b"variant", old_v)
        hip.FLAGSTATS_hip_set(b"blocks_per_cu", old_b)
        hip.FLAGSTATS_hip_set(b"fuse", old_f)
        hip.FLAGSTATS_hip_set(b"epilogue", old_e)
        buf.free()


def test_values_concentrated_on_single_categories(hip):
    """Arrays made of ONE value each: every LUT entry of the front end is hit in isolation, for every
    value of the 10 FLAG bits the rule reads (bits 4,5,12-15 are ignored, libflagstats.h:118-142)."""
    import oracle
    from libflagstats_amd import device
    rel = [0, 1, 2, 3, 6, 7, 8, 9, 10, 11]
    n = 4099
    buf = device.DeviceFlags(n)
    rs = np.random.RandomState(1)
    for code in range(1024):
        v = sum(((code >> i) & 1) << b for i, b in enumerate(rel)) | (int(rs.randint(0, 4)) << 4) | (int(rs.randint(0, 16)) << 12)
        a = np.full(n, v, dtype=np.uint16)
        buf.upload(a)
        got = buf.count()
        one = oracle.flagstat_c(a[:1])
        assert np.array_equal(got, one * np.uint64(n)), hex(v)
    buf.free()


@pytest.mark.parametrize("form", ["atomic", "k2", "ticket"])
def test_back_to_back_launches_same_workspace(hip, form):
    """Many launches of very different sizes on one stream and one workspace, no host sync between them,
    in every finalisation form: K1's atomic epilogue (default), partials + K2, and -- tuning build only --
    the r01 ticket form, whose last-arriving workgroup finalises and re-arms the ticket for the next
    launch.  The accumulated device counters must equal the sum of the oracle's, and a final store-form
    call must overwrite."""
    import torch

    import oracle
    from libflagstats_amd import _lib, device
    if form == "ticket" and not hip.FLAGSTATS_hip_get(b"tuning_build"):
        assert hip.FLAGSTATS_hip_set(b"fuse", 1) != 0 and b"TUNING=1" in hip.FLAGSTATS_hip_last_error()
        return
    old = hip.FLAGSTATS_hip_get(b"fuse")
    old_e = hip.FLAGSTATS_hip_get(b"epilogue")
    _lib.check(hip.FLAGSTATS_hip_set(b"fuse", 1 if form == "ticket" else 0), "fuse")
    _lib.check(hip.FLAGSTATS_hip_set(b"epilogue", 1 if form == "atomic" else 0), "epilogue")
    try:
        n = 50_000_000
        t = torch.empty(n, dtype=torch.int16, device="cuda:0")
        device.generate_torch(t, device.GEN_UNIFORM, seed=4242, mask=0xFFFF)
        host = oracle.generate(oracle.GEN_UNIFORM, 4242, 0xFFFF, 0, n)
        rs = np.random.RandomState(9)
        out = torch.zeros(32, dtype=torch.int64, device="cuda:0")
        want = np.zeros(32, dtype=np.uint64)
        for i in range(300):                      # no host sync between launches
            a = int(rs.randint(0, n - 1))
            m = int(rs.choice([1, 7, 300, 16384, 70000, 3_000_000, 20_000_000]))
            m = min(m, n - a)
            device.count_torch(t[a:a + m], out)
            want += oracle.flagstat_hist(host[a:a + m])
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
        device.count_torch(t[5:1005], out, store=True)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oracle.flagstat_hist(host[5:1005]))
    finally:
        hip.FLAGSTATS_hip_set(b"fuse", old)
        hip.FLAGSTATS_hip_set(b"epilogue", old_e)
