import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# tests of the MEASUREMENT build (make tuning) are not part of the product's suite: neither `-m gpu` nor `-m "not gpu"` sees them
collect_ignore_glob = [] if os.environ.get("FLAGSTATS_TUNING_TESTS") else ["tuning/*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "tuning: needs a real MI355X AND the measurement build of the library (tests/tuning/)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def case_input(case):
    """Rebuild the input of a tests/golden/random_cases.json entry from its recipe."""
    a = np.random.RandomState(case["seed"]).randint(0, case["hi"], case["n"] + case["skip"]).astype(np.uint16)
    a = a[case["skip"]:]
    if "input" in case:
        assert [int(v) for v in a] == case["input"], "numpy RandomState drifted from the committed fixture"
    return a


def inmemory_input(n):
    """benchmark/inmemory.cpp:108-116 input: libstdc++ uniform_int_distribution<uint16_t>(0,4095)
    over mt19937 seeded with 0 (recipe recorded in tests/golden/inmemory_mt19937.json)."""
    bg = np.random.MT19937()
    bg._legacy_seeding(0)  # init_genrand(0) == std::mt19937::seed(0)
    raw = bg.random_raw(n)
    # libstdc++ (GCC >= 11) draws with Lemire's method: (raw * 4096) >> 32, and a
    # power-of-two range never rejects
    return (raw >> 20).astype(np.uint16)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.load_c()
    return oracle


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library, initialised on cuda:0.  GPU tests only."""
    from libflagstats_amd import _lib
    lib = _lib.lib()
    assert lib.FLAGSTATS_hip_available() == 1, "no gfx950 device visible"
    _lib.check(lib.FLAGSTATS_hip_init(0), "FLAGSTATS_hip_init")
    # the tests call the reference-shaped entry points directly and check their return codes: a failure must fail
    # the test, not abort() the test runner (the library's default for callers that ignore the return value)
    lib.FLAGSTATS_hip_set(b"on_error", 0)
    return lib
