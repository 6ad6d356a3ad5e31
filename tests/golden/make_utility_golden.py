#!/usr/bin/env python3
"""Build-container only: runs the reference's own `utility` (benchmark/utility.cpp, compiled unmodified
by oracle/Makefile into oracle/_ref/utility_ref) on seeded / hand-written FLAG text and records
input -> output as data in tests/golden/utility_cases.json."""
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "utility_ref")


def run(text: bytes):
    out = subprocess.run([EXE], input=text, capture_output=True, check=True).stdout
    return [int(v) for v in np.frombuffer(out, dtype=np.uint16)]


def main():
    rs = np.random.RandomState(3)
    cases = []

    def add(name, text):
        cases.append({"name": name, "text": text.decode("latin-1"), "values": run(text)})

    add("empty", b"")
    add("one_no_newline", b"99")
    add("one_newline", b"99\n")
    add("trailing_blank_lines", b"147\n\n\n")
    add("crlf", b"83\r\n163\r\n")
    add("leading_space_sign_junk", b"  99\n\t+147\n-1\nabc\n12abc\n 0x10\n")
    add("wraps_to_16_bits", b"65535\n65536\n70000\n131071\n-65535\n")
    add("samtools_like", b"\n".join(str(int(v)).encode() for v in rs.choice([99, 147, 83, 163, 2113, 77, 1024 + 99], 500)) + b"\n")
    add("full_range", b"\n".join(str(int(v)).encode() for v in rs.randint(0, 65536, 400)))
    with open(os.path.join(HERE, "utility_cases.json"), "w") as f:
        json.dump({"generator": "oracle/_ref/utility_ref (benchmark/utility.cpp, unmodified) via tests/golden/make_utility_golden.py",
                   "cases": cases}, f, indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
