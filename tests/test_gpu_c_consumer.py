"""GPU: a plain C consumer of the reference API (tests/consumer_shim.c: FLAGSTATS_get_function +
FLAGSTATS_u16, compiled against the header shim, linked with libflagstats_hip.so) run as its own
process on the MI355X; its printed counters are checked against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_consumer_process(hip, tmp_path):
    import oracle
    exe = str(tmp_path / "consumer")
    libdir = os.path.join(ROOT, "libflagstats_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "consumer_shim.c"),
                    "-L", libdir, "-lflagstats_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe],
                   check=True)
    for n in (1, 1000, 512000, 3_000_001):
        r = subprocess.run([exe, str(n)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"rc=0 rc2=0 unmapped=(\d+) qcfail=(\d+) dup=(\d+)", r.stdout)
        assert m, r.stdout
        i = np.arange(n, dtype=np.uint64)
        flags = ((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)   # consumer_shim.c's input
        want = oracle.flagstat_hist(flags.astype(np.uint16)) * np.uint64(2)                # the consumer counts twice (+=)
        assert [int(v) for v in m.groups()] == [int(want[2]), int(want[25]), int(want[10])], n
