"""CPU: row f3's text -> uint16 converter (FLAGSTATS_text_to_u16, the counterpart of the reference's
`utility`, benchmark/utility.cpp:9-16) against input/output pairs produced by the reference's own
binary (tests/golden/utility_cases.json, tests/golden/make_utility_golden.py).  Host code: no GPU."""
import ctypes
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

from libflagstats_amd import _lib, textio


@pytest.mark.parametrize("case", load_golden("utility_cases.json")["cases"], ids=lambda c: c["name"])
def test_matches_reference_utility(case):
    text = case["text"].encode("latin-1")
    got = textio.flags_from_text(text)
    assert got.dtype == np.uint16 and got.tolist() == case["values"]


def test_capacity_is_enforced_loudly():
    lib = _lib.lib()
    out = np.zeros(2, dtype=np.uint16)
    assert lib.FLAGSTATS_text_to_u16(b"1\n2\n3\n", 6, out.ctypes.data, 2) < 0
    assert b"more lines" in lib.FLAGSTATS_hip_last_error()
    assert lib.FLAGSTATS_text_to_u16(b"1\n2\n", 4, out.ctypes.data, 2) == 2 and out.tolist() == [1, 2]
    assert lib.FLAGSTATS_text_count_lines(b"1\n2", 3) == 2 and lib.FLAGSTATS_text_count_lines(b"", 0) == 0


def test_cli_is_a_drop_in_for_the_reference_pipeline():
    text = b"99\n147\n83\n163\n"
    out = subprocess.run([sys.executable, ROOT + "/tools/utility.py"], input=text, capture_output=True, check=True).stdout
    assert np.frombuffer(out, dtype=np.uint16).tolist() == [99, 147, 83, 163]
